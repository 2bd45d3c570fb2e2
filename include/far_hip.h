/* far_hip.h -- C ABI of libfar_hip.so: the MI355X (gfx950) kernels behind FAR's pose-estimation hot path.
 *
 * The reference (crockwell/far, mp3d_loftr) has no native code and no FFI: its boundary is the Python
 * nn.Module / data-dict API.  This header is the boundary a maintainer binds with
 * ctypes from those modules; every entry point names the reference code it replaces.
 * Citations are relative to the reference root (mp3d_loftr/...).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host; no allocation inside the library:
 *     the caller passes workspaces sized by the matching *_workspace_bytes();
 *   - every function enqueues on `stream` and returns immediately: 0 = enqueued, negative errno otherwise
 *     (-22 invalid argument, -5 launch failure); nothing throws, nothing prints;
 *   - process-global state is limited to (a) per-device one-time kernel attribute setup and (b) the speed-only
 *     A/B knobs of far_set_tuning() (atomics; they never change results); the side streams of far_stream_fork /
 *     far_stream_join belong to the calling host thread (per device); the usual deployment is one process per GPU,
 *     but launching on several devices, or from several threads, of one process is supported;
 *   - "Z" is a flat batch (image pairs, or pairs x heads x directions for the head).
 */
#ifndef FAR_HIP_H
#define FAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* far_stream_t; /* == hipStream_t */

/* ABI version of this header; bumped when a signature changes (2: activation exponent / overflow flag of K9, K13, K14; 3: the
 * far_wino_* / far_conv3x3_wino_f32 entry points, 16 tuning keys; 4: far_upsample2x_bwd_f32, far_fine_scatter_det_f32, far_bn_train_*, far_adamw_*; 5: far_linear_kv_f16s, far_linear_q_apply_f16s,
 * far_linear_gather_f16s, far_linear_attention_apply_f32, far_prior_from_pose_f32; 6: far_ransac_f64, far_eightpoint_f64, far_decompose_essential_f64, far_build_id; 7: far_emm_pv_f16, far_attn_block_f16, far_mlp_fused_f16; far_linear_kv_f16s / far_linear_q_apply_f16s accept split = 0).  far_amd/_lib.py refuses a library whose version differs. */
int far_abi_version(void);
/* Id of the sources the library was built from: sha256/16 over far_amd/csrc/* and the compiler flags (far_amd/build.py
 * source_id()).  far_amd/_lib.py refuses a library whose id differs from the sources it sits next to. */
const char* far_build_id(void);
/* hipError_t of the most recent failed launch on the calling thread (0 = none): detail behind a -5 return. */
int far_last_hip_error(void);
/* Tuning knobs for A/B experiments; they change speed only, never results.  key 0 = bit mask of kernels using
 * wave-slot issue-priority staggering (1 k_stats, 2 k_match, 4 k_emm_pv). */
int far_set_tuning(int key, int value);

/* Side streams (FAR_SIDE_STREAMS = 4 per device, created on first use): independent launches of a batch-1 training step -- a
 * layer's weight gradient next to its input gradient, its q / k / v projections -- each fill a fraction of the CUs and overlap
 * when issued on different streams; the streams belong to the calling host thread.  far_stream_fork(main, i): side stream i waits for what `main` holds so far; returns the
 * stream to launch on (NULL on failure).  far_stream_join(main, i): `main` waits for side stream i.  Buffers used on a side
 * stream must stay allocated until the join. */
void* far_stream_fork(far_stream_t main, int i);
int far_stream_join(far_stream_t main, int i);

/* Measurement aid (no reference counterpart): the f16 matrix-pipe rate this part SUSTAINS under dense
 * v_mfma_f32_32x32x16_f16 issue with the register / LDS footprint of K9 / K1 / K2 (2 x 4 accumulator tiles per wave,
 * 4 waves per workgroup, 2 workgroups per CU), pseudo-random operands.  mode 0: operands in registers; mode 1: the six
 * operand fragments re-read from LDS every step.  Launches rounds x 2 x CU-count workgroups, `iters` steps of 8 MFMAs per
 * wave; *flops_out (host memory, may be NULL) = flops executed by the launch.  bench.py times it with events and reports
 * roofline.sustained_peak; `sink` is one device float that is never written. */
int far_mfma_probe_f16(int mode, int iters, int rounds, float* sink, double* flops_out, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K1  coarse matcher: all-pairs correlation + dual-softmax + mutual-NN selection
 * replaces src/loftr/utils/coarse_matching.py:86-147 (CoarseMatching.forward, dual_softmax branch)
 *      and src/loftr/utils/coarse_matching.py:149-265 (get_coarse_match, eval path) incl. mask_border :8-25
 * ------------------------------------------------------------------------------------------------- */

/* Bytes of workspace for Z problems of L x S tokens (shared by the three functions below). */
size_t far_dual_softmax_workspace_bytes(int Z, int L, int S);

/* Row / column softmax statistics of  sim[z,i,j] = ((f0[z,i,:]/feat_div) . (f1[z,j,:]/feat_div)) / sim_div * sim_mul
 *   f0 [Z][L][C], f1 [Z][S][C] fp32 contiguous, C % 32 == 0
 *   mask0 [Z][L], mask1 [Z][S] optional uint8 (0 = padded): masked pairs get sim = -1e9 (coarse_matching.py:114-117)
 *   rowstat_out [Z][L][2], colstat_out [Z][S][2] optional: (max, sum exp(sim - max)); always also left in ws. */
int far_dual_softmax_stats_f32(const float* f0, const float* f1, int Z, int L, int S, int C,
                               float feat_div, float sim_div, float sim_mul,
                               const uint8_t* mask0, const uint8_t* mask1,
                               float* rowstat_out, float* colstat_out, void* ws, far_stream_t stream);

/* Full matcher.  f0/f1 as above with feat_div = sqrt(C), sim_div = temperature (coarse_matching.py:104-113).
 *   thr, border: match_coarse.thr / border_rm; (h0,w0),(h1,w1): coarse grids, h0*w0 == L, h1*w1 == S
 *   cell_scale: hw0_i[0] / hw0_c[0] (coarse_matching.py:246); scale0/scale1 optional [Z][2] (x,y) per-pair image scales
 *   valid_hw optional [Z][4] = valid (h0,w0,h1,w1) extents for mask_border_with_padding (:28-43)
 *   conf_out optional [Z][L][S]: data['conf_matrix'] (needed by the loss / plotting only)
 *   outputs sized for the worst case Z*L: b_ids,i_ids,j_ids int64; mconf fp32; mkpts0_c/mkpts1_c [.,2] fp32,
 *   ordered by (b, i) exactly like torch.where on the reference's mask; counts_out optional [Z]; *total_out = M. */
int far_coarse_match_f32(const float* f0, const float* f1, int Z, int L, int S, int C,
                         float temperature, float thr, int border, int h0, int w0, int h1, int w1,
                         float cell_scale, const uint8_t* mask0, const uint8_t* mask1,
                         const int* valid_hw, const float* scale0, const float* scale1,
                         float* conf_out, int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf,
                         float* mkpts0_c, float* mkpts1_c, int* counts_out, int* total_out,
                         void* ws, far_stream_t stream);

/* bf16-input variant of the matcher: features are rounded to bf16 (RNE) once, the contraction runs on the bf16
 * matrix core with fp32 accumulation, everything after the dot product is the fp32 code of the exact variant.
 * Same arguments / outputs as far_coarse_match_f32; C == 256 (the coarse feature width); workspace from the query below. */
size_t far_coarse_match_bf16_workspace_bytes(int Z, int L, int S, int C);
int far_coarse_match_bf16(const float* f0, const float* f1, int Z, int L, int S, int C,
                          float temperature, float thr, int border, int h0, int w0, int h1, int w1,
                          float cell_scale, const uint8_t* mask0, const uint8_t* mask1,
                          const int* valid_hw, const float* scale0, const float* scale1,
                          float* conf_out, int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf,
                          float* mkpts0_c, float* mkpts1_c, int* counts_out, int* total_out,
                          void* ws, far_stream_t stream);

/* Split-fp16 variant of far_coarse_match_f32 (fp32 features, hi + lo fp16 operand pairs on the f16 matrix cores,
 * fp32 accumulation: an fp32-grade similarity at 16/3 of the exact-f32 MFMA rate; one exp per score; conf_matrix, if
 * requested, written with 16-byte stores).  Same arguments and outputs; C must be 256.
 * overflow (this and the other *_f16s entry points of K1 / K2; device int or NULL): OR-ed with 1 when a feature is beyond
 * the range of the 2^4-scaled fp16 split (|x| > 4094, or NaN) -- the scores are then NaN and NO match is reported, so a
 * caller must look at the flag (far_amd/loftr does, and re-runs on far_coarse_match_f32, which has no such limit). */
size_t far_coarse_match_f16s_workspace_bytes(int Z, int L, int S, int C);
int far_coarse_match_f16s(const float* f0, const float* f1, int Z, int L, int S, int C,
                          float temperature, float thr, int border, int h0, int w0, int h1, int w1,
                          float cell_scale, const uint8_t* mask0, const uint8_t* mask1,
                          const int* valid_hw, const float* scale0, const float* scale1,
                          float* conf_out, int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf,
                          float* mkpts0_c, float* mkpts1_c, int* counts_out, int* total_out,
                          void* ws, int* overflow, far_stream_t stream);

/* data['conf_matrix'] alone -- CoarseMatching.forward's dual-softmax matrix (coarse_matching.py:108-118), which only the
 * dense coarse loss and plotting consume (loftr_loss.py:307-311) -- written at HBM speed: statistics from the
 * split-precision passes above (fp32-grade), the (Z, L, S) matrix from plain-fp16 scores (one MFMA per 16 channels,
 * one exp per score, full-line stores), and every entry that can exceed 2^-12 -- about one per row, the only ones an
 * fp16 operand error can move by more than 1e-5 -- rewritten from the split-precision score the statistics pass saw.
 *   stages        bit 0: operand planes + statistics; bit 1: write the matrix (callers that keep `ws` may split them)
 *   fix_info_out  optional 2 device ints: entries rewritten exactly, entries that did not fit their slot list (> 0:
 *                 those kept their fp16-operand value, relative error ~1e-3; use far_coarse_match_f16s then)
 *   ws            far_coarse_match_f16s_workspace_bytes(Z, L, S, C) bytes; C must be 256. */
int far_conf_matrix_f16s(const float* f0, const float* f1, int Z, int L, int S, int C, float temperature,
                         const uint8_t* mask0, const uint8_t* mask1, int stages, float* conf_out, int* fix_info_out,
                         void* ws, int* overflow, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K1 on the training path: sparse coarse supervision without conf_matrix / conf_matrix_gt
 * replaces, for match_type 'dual_softmax' + sparse_spvs + focal loss (the FAR training configuration):
 *   src/loftr/utils/coarse_matching.py:108-118   conf_matrix = softmax(sim, 1) * softmax(sim, 2)     (92 MB / pair)
 *   src/loftr/utils/supervision.py:113-137       conf_matrix_gt                                       (92 MB / pair)
 *   src/losses/loftr_loss.py:86-91               pos_conf = conf[pos_mask]   -- the ONLY read of the dense matrix
 *   and autograd's backward through the two softmaxes and the correlation
 * positions (pb, pi, pj)[M] int64 = spv_b_ids / spv_i_ids / spv_j_ids.  C must be 256.
 * ws: far_coarse_train_workspace_bytes(Z, L, S, C) bytes, kept untouched between the forward and the backward call. */
size_t far_coarse_train_workspace_bytes(int Z, int L, int S, int C);
/* forward: p_out[k] = conf_matrix[pb[k], pi[k], pj[k]] (fp32-grade statistics; float64 dot product at the positions). */
int far_coarse_pos_conf_f16s(const float* f0, const float* f1, int Z, int L, int S, int C, float temperature,
                             const int64_t* pb, const int64_t* pi, const int64_t* pj, int M, float* p_out, void* ws,
                             int* overflow, far_stream_t stream);
/* backward: w[k] = dL/dp_k * p_k.  df0 (Z, L, C), df1 (Z, S, C) are overwritten with dL/dfeat_c0, dL/dfeat_c1.
 * Dense part on the f16 matrix cores (recomputed score tiles, G = u R + v C fed from registers into the second MFMA),
 * plain fp16 operands / fp32 accumulation: gradient-grade (~1e-3 relative), not parity-grade. */
int far_coarse_pos_conf_bwd_f16(const float* f0, const float* f1, int Z, int L, int S, int C, float temperature,
                                const int64_t* pb, const int64_t* pi, const int64_t* pj, int M, const float* w,
                                float* df0, float* df1, void* ws, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K2  EMM head: bilinear dual-softmax attention  F = v~^T (softmax_row(s) * softmax_col(s)) v~
 * replaces src/loftr/loftr_module/transformer.py:275-292 (CrossAttention.forward), one call per direction
 * ------------------------------------------------------------------------------------------------- */

/* T[z] = P[z] @ [v[z] | pos]  (Z, N, 70), P = softmax over keys * softmax over queries of s = (q k^T) * scale.
 *   q, k, v [Z][N][64] fp32 contiguous (Z = pairs x heads); pos [N][6] fp32, shared (transformer.py:183-248)
 *   rowstat/colstat [Z][N][2]: from far_dual_softmax_stats_f32(q, k, feat_div=1, sim_div=1, sim_mul=scale)
 *   The caller finishes F[z] = [v|pos]^T T[z] (70 x N x 70) with a library GEMM. */
int far_emm_pv_f32(const float* q, const float* k, const float* v, const float* pos, int Z, int N, int D,
                   float scale, const float* rowstat, const float* colstat, float* T_out, far_stream_t stream);

/* The whole K2 operator (softmax statistics included) on the f16 matrix cores with split-precision operands
 * (fp32 tensors, hi + lo fp16 pairs, three MFMAs per product, fp32 accumulation: fp32-grade).  Same inputs / output
 * as far_emm_pv_f32, except that q, k, v may be strided per problem: z = p * heads + hh starts at
 * ptr + hh * head_stride + p' * prob_stride (floats), p' = p for k, v and (p + q_rot) mod (Z / heads) for q -- the
 * layout in which the head's fused q | k | v projection (far_conv_nhwc_f32 with out_planes = 12) leaves them, so
 * the reference's reshape/permute (transformer.py:270-274) needs no copy.  Contiguous [Z][N][64]: heads = 1,
 * prob_stride = 64 N, q_rot = 0.  ws: far_emm_pv_f16s_workspace_bytes(Z, N) bytes of scratch. */
size_t far_emm_pv_f16s_workspace_bytes(int Z, int N);
int far_emm_pv_f16s(const float* q, const float* k, const float* v, const float* pos, int Z, int N, int D, float scale,
                    int heads, long head_stride, long prob_stride, int q_rot, void* ws, float* T_out, int* overflow,
                    far_stream_t stream);
/* The 16-bit-operand form of the same operator (the precision class BASELINE configs[1] runs the reference in: fp16
 * autocast, transformer.py:275-292 under torch.autocast): plain fp16 operands, one MFMA product per tile, fp32
 * accumulation and fp32 softmax statistics.  Arguments, workspace and overflow flag exactly as far_emm_pv_f16s. */
int far_emm_pv_f16(const float* q, const float* k, const float* v, const float* pos, int Z, int N, int D, float scale,
                   int heads, long head_stride, long prob_stride, int q_rot, void* ws, float* T_out, int* overflow,
                   far_stream_t stream);
/* the statistics of the last far_emm_pv_f16s call on `ws`: rowstat / colstat [Z][N][2] = (max, sum) in the log2 domain */
int far_emm_pv_f16s_copy_stats(const void* ws, int Z, int N, float* rowstat_out, float* colstat_out, far_stream_t stream);

/* K2 backward (training path): dq, dk [Z][N][64] of F = vt^T P vt (transformer.py:275-292) -- what the reference's
 * autograd derives through two (B, 4, 4800, 4800) softmax tensors per direction, here from recomputed 32 x 32 tiles.
 *   vt = [v | pos] and A = vt dF: [Z][N][70];  u = rowdot(A, P vt), vw = rowdot(vt dF^T, P^T vt): [Z][N]
 *   rowstat / colstat: far_emm_pv_f16s_copy_stats of the forward.  A, u, vw may share a power-of-two scale.
 * Plain fp16 operands, fp32 accumulation: gradient-grade (~1e-3).  dv~ = (P vt) dF^T + (P^T vt) dF is the caller's. */
size_t far_emm_bwd_workspace_bytes(int Z, int N);
int far_emm_bwd_f16(const float* q, const float* k, const float* vt, const float* A, const float* u, const float* vw,
                    const float* rowstat, const float* colstat, int Z, int N, float scale, float* dq, float* dk,
                    void* ws, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K3  fine level: window gather + sub-pixel expectation
 * replaces src/loftr/loftr_module/fine_preprocess.py:40-47 (F.unfold + [b_ids, i_ids] gather)
 *      and src/loftr/utils/fine_matching.py:43-54, :64-76 (FineMatching.forward / get_fine_match)
 * ------------------------------------------------------------------------------------------------- */

/* out[m][ky*W+kx][c] = feat[b_ids[m]][c][y0*stride - W/2 + ky][x0*stride - W/2 + kx] (0 outside),
 * (y0,x0) = divmod(cell_ids[m], wc).  feat is addressed with ELEMENT strides (sn,sc,sh,sw) so NCHW and
 * channels_last maps both work; logical shape [*, C, Hf, Wf]; out [M][W*W][C].  M == 0 is a no-op. */
int far_fine_gather_f32(const float* feat, long sn, long sc, long sh, long sw, int C, int Hf, int Wf,
                        const int64_t* b_ids, const int64_t* cell_ids, int wc, int W, int stride, int M,
                        float* out, far_stream_t stream);

/* feat0/feat1 [M][W*W][C] (after the fine transformer).  expec_f [M][3] = (E[x], E[y], std) in normalised
 * window coordinates; mkpts1_f [M][2] = mkpts1_c + E[xy] * win_scale (* scale1[b_ids[m]] when scale1 != NULL),
 * win_scale = (W // 2) * (hw0_i[0] / hw0_f[0]). */
/* backward of far_fine_gather_f32 (training): dfeat (feat's strides and shape) += dout [M][W*W][C] at the gathered positions;
 * fp32 atomics (overlapping windows, cells sampled with replacement: coarse_matching.py:216-229). */
int far_fine_scatter_f32(const float* dout, long sn, long sc, long sh, long sw, int C, int Hf, int Wf,
                         const int64_t* b_ids, const int64_t* cell_ids, int wc, int W, int stride, int M,
                         float* dfeat, far_stream_t stream);
/* The same backward with a fixed summation order (run-to-run bit-identical; far_fine_scatter_f32 adds with fp32 atomics):
 * order = the match indices sorted stably by b_ids * (hc * wc) + cell_ids, start = Z * hc * wc + 1 int32 group offsets in that
 * order.  One wave per fine-map pixel sums the windows that cover it, cells in ascending (row, column) order. */
int far_fine_scatter_det_f32(const float* dout, long sn, long sc, long sh, long sw, int C, int Hf, int Wf, const int64_t* order,
                             const int* start, int Z, int hc, int wc, int W, int stride, int M, float* dfeat, far_stream_t stream);
int far_fine_expect_f32(const float* feat0, const float* feat1, int M, int W, int C, const float* mkpts1_c,
                        float win_scale, const float* scale1, const int64_t* b_ids, float* expec_f,
                        float* mkpts1_f, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K5  linear attention core of LoFTREncoderLayer
 * replaces src/loftr/loftr_module/linear_attention.py:31-50 (LinearAttention.forward)
 * ------------------------------------------------------------------------------------------------- */
size_t far_linear_attention_workspace_bytes(int N, int S, int H, int D);

/* q [N][L][H*D], k, v [N][S][H*D] raw projections (the elu+1 feature map is applied inside);
 * q_mask [N][L], kv_mask [N][S] optional uint8; eps = 1e-6; D in {16, 32}; out [N][L][H*D]. */
int far_linear_attention_f32(const float* q, const float* k, const float* v, int N, int L, int S, int H, int D,
                             const uint8_t* q_mask, const uint8_t* kv_mask, float eps, float* out, void* ws,
                             far_stream_t stream);

/* The second half alone: out from q and the per-image state kv [N][H*32][33] (K'^T (V / S) per head; 33rd column: the sum of K')
 * that far_linear_kv_f16s leaves (D = 32) -- the launch far_linear_attention_f32 ends with (linear_attention.py:46-50). */
int far_linear_attention_apply_f32(const float* q, const float* kv, int N, int L, int S, int H, const uint8_t* q_mask, float eps,
                                   float* out, far_stream_t stream);

/* K5 backward (training path): gradients of far_linear_attention_f32 w.r.t. the raw projections q, k, v given
 * g = dL/dout -- what autograd derives from linear_attention.py:31-50 in the reference.  Token-parallel kernels with the
 * head's D x D matrices in LDS; the token reductions (dKV, dksum) in fixed order (deterministic). */
size_t far_linear_attention_bwd_workspace_bytes(int N, int L, int S, int H, int D);
int far_linear_attention_bwd_f32(const float* q, const float* k, const float* v, const float* g, int N, int L, int S,
                                 int H, int D, const uint8_t* q_mask, const uint8_t* kv_mask, float eps,
                                 float* dq, float* dk, float* dv, void* ws, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K6  LayerNorm (+ fused residual) of the encoder layers and the head
 * replaces src/loftr/loftr_module/transformer.py:61, :65-67 (norm1; norm2 + `x + message`), :342, :346, :426
 * ------------------------------------------------------------------------------------------------- */

/* y[r][:] = LayerNorm(x[r][:]; eps) * gamma + beta (+ res[r][:] when res != NULL).  x, res, y [rows][C] fp32. */
int far_layernorm_f32(const float* x, const float* gamma, const float* beta, const float* res, long rows, int C,
                      float eps, float* y, far_stream_t stream);
/* Backward of the same normalisation (autograd of nn.LayerNorm at transformer.py:61, 65-67, training): dx [rows][C],
 * dgamma [C], dbeta [C] from x, gamma, dy; the statistics are recomputed as the forward forms them, dgamma / dbeta are summed
 * in a fixed order (deterministic).  C % 4 == 0, C <= 1024 (far_layernorm_bwd_ws_bytes() = 0 otherwise); ws: that many bytes of
 * device scratch.  A fused residual needs nothing here: its gradient is dy itself. */
long far_layernorm_bwd_ws_bytes(long rows, int C);
int far_layernorm_bwd_f32(const float* x, const float* gamma, const float* dy, long rows, int C, float eps, float* dx,
                          float* dgamma, float* dbeta, void* ws, long ws_bytes, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Training driver of one LoFTR encoder layer: LoFTREncoderLayer.forward under autograd
 * (src/loftr/loftr_module/transformer.py:44-67) as one call for the forward and one for the backward.
 * No arithmetic of its own: K9 / K5 / K6 / K16 / far_grad_scale_f32 launched in sequence (the k / v projections and every weight
 * gradient on side streams when `overlap`), so that a layer costs the host two calls instead of ~50 launches' worth of
 * interpreter time.  x (bs, L, C), source (bs, S, C) fp32 contiguous (self_attn: source ignored, S = L);
 * C % 4 == 0, C <= 512, C / nhead in {16, 32}; bias-free Linear layers.  img[i] / img_scale[i]: the K9 image and epilogue scale
 * vector of q, k, v, merge, mlp[0], mlp[2] (far_conv_pack_view_scaled_f32); imgT / imgT_scale: the transposed (dgrad) images.
 * Buffers are the caller's: `saved` (forward -> backward), `grads` (layout by far_enc_layer_grads_offsets: dx, ds (-1 for
 * self-attention), dW x 6 in the weights' torch layouts, dgamma1, dbeta1, dgamma2, dbeta2), scratch `ws`.
 * --------------------------------------------------------------------------------------------------- */
typedef struct far_enc_layer {
    long bs, L, S;
    int C, nhead, self_attn, split, act_exp, overlap;
    float eps1, eps2, attn_eps;
    const void* img[6];
    const float* img_scale[6];
    const void* imgT[6];
    const float* imgT_scale[6];
    const float *g1, *b1, *g2, *b2;
    int* overflow;
} far_enc_layer;
long far_enc_layer_saved_floats(const far_enc_layer* d);
long far_enc_layer_grads_floats(const far_enc_layer* d);
long far_enc_layer_fwd_ws_bytes(const far_enc_layer* d);
long far_enc_layer_bwd_ws_bytes(const far_enc_layer* d);
int far_enc_layer_grads_offsets(const far_enc_layer* d, long* out12);
int far_enc_layer_fwd(const far_enc_layer* d, const float* x, const float* source, float* saved, float* y, void* ws, long ws_bytes,
                      far_stream_t stream);
int far_enc_layer_bwd(const far_enc_layer* d, const float* x, const float* source, const float* saved, const float* gy, float* grads,
                      void* ws, long ws_bytes, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K13  the MLP block of a LoFTR encoder layer at d_model = 128 (the fine-level transformer) in one launch
 * replaces src/loftr/loftr_module/transformer.py:64-67:  x + norm2(mlp(cat[x, message]))  with
 *          mlp = Linear(2d, 2d, no bias) -> ReLU -> Linear(2d, d, no bias); the hidden tensor never reaches memory.
 *   x, msg [R][128] fp32; packed = the image far_amd/ops/fine.py:PackedMlp builds (far_mlp_fused_packed_bytes bytes: W0 scaled by
 *   2^e0, W2 by 2^e2, fp16 hi / lo planes in execution order); hscale = 2^-e0, oscale = 2^-(e2 + 4); gamma, beta [128], eps:
 *   norm2.  out [R][128] must not alias x or msg.  Arithmetic as K9 (three f16 MFMAs per fp32-grade product).
 * --------------------------------------------------------------------------------------------------- */
/* K14  the attention block of a LoFTR encoder layer at d_model = 128 on sequences of <= 32 tokens (the fine-level windows)
 * replaces src/loftr/loftr_module/transformer.py:51-61 with linear_attention.py:31-50:
 *          norm1(merge(LinearAttention(q_proj(x), k_proj(src), v_proj(src))))      8 heads of 16 channels
 *   x [nwin][L][128], src [nwin][S][128] fp32 (L, S <= 32); packed = the image far_amd/ops/fine.py:PackedAttn builds
 *   (far_attn_block_packed_bytes bytes); scale_k / _v / _q / _m = 2^-(w_exp + 4) of the four weight tensors; attn_eps: the
 *   1e-6 of LinearAttention; gamma, beta [128], ln_eps: norm1.  out [nwin][L][128] must not alias x or src.
 *   overflow (both kernels; device int or NULL): OR-ed with 1 when an input or an intermediate left the range of the
 *   2^4-scaled fp16 split (|value| > 4094; for K14 that includes the per-head K'^T V / S sums and the message) -- the
 *   result is then inf / NaN, never silently wrong; far_amd/loftr re-runs the layer on K9 + K5 with a lower exponent. */
size_t far_attn_block_packed_bytes(int d_model);
int far_attn_block_f16s(const float* x, const float* src, const void* packed, long nwin, int L, int S, int d_model, int heads,
                        float scale_k, float scale_v, float scale_q, float scale_m, float attn_eps, const float* gamma,
                        const float* beta, float ln_eps, float* out, int* overflow, far_stream_t stream);
size_t far_mlp_fused_packed_bytes(int d_model);
int far_mlp_fused_f16s(const float* x, const float* msg, const void* packed, long R, int d_model, float hscale, float oscale,
                       const float* gamma, const float* beta, float eps, float* out, int* overflow, far_stream_t stream);
/* K14 / K13 on plain fp16 operands (one MFMA per product, fp32 accumulation; the 16-bit-operand class the reference runs in under
 * autocast, LoFTR.set_precision('fp16')): same arguments, same packed images, same overflow flag as the _f16s entry points. */
int far_attn_block_f16(const float* x, const float* src, const void* packed, long nwin, int L, int S, int d_model, int heads,
                       float scale_k, float scale_v, float scale_q, float scale_m, float attn_eps, const float* gamma,
                       const float* beta, float ln_eps, float* out, int* overflow, far_stream_t stream);
int far_mlp_fused_f16(const float* x, const float* msg, const void* packed, long R, int d_model, float hscale, float oscale,
                      const float* gamma, const float* beta, float eps, float* out, int* overflow, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K7 / K8  backbone epilogues (inference): folded BatchNorm + residual + activation; FPN upsample + add
 * replaces the elementwise passes of src/loftr/backbone/resnet_fpn.py:32-40, :80-91, :103, :110-116
 * ------------------------------------------------------------------------------------------------- */

/* y = act(x * scale[c] + shift[c] (+ res)) on activations [N][C][H][W]: nhwc = 0 contiguous NCHW (HW % 4 == 0),
 * nhwc = 1 channels_last memory (C % 4 == 0).  act: 0 none, 1 ReLU, 2 LeakyReLU(slope).  y may alias x. */
int far_affine_act_f32(const float* x, const float* scale, const float* shift, const float* res, long N, int C,
                       long HW, int nhwc, int act, float slope, float* y, far_stream_t stream);

/* out = hi + bilinear 2x upsample (align_corners = True) of lo; lo [N][C][h][w], hi/out [N][C][2h][2w], both in the
 * layout selected by nhwc (0: NCHW, w even; 1: channels_last, C % 4 == 0). */
int far_upsample2x_add_f32(const float* lo, const float* hi, int N, int h, int w, int C, int nhwc, float* out,
                           far_stream_t stream);

/* K20: AdamW over all parameter tensors of a model in one launch (adamw_f32.hip) -- torch.optim.AdamW's arithmetic in its order
 * (src/optimizers/__init__.py:5-16): p *= 1 - lr wd; m += (1 - beta1)(g - m); v = beta2 v + (1 - beta2) g g;
 * p -= (lr / bc1) m / (sqrt(v) / bc2_sqrt + eps).  table: far_adamw_table_bytes(n, nblocks) bytes on the device: n rows
 * { float* p; const float* g; float* m; float* v; long n; } then nblocks int2 { tensor, chunk of 4096 elements }; a row with g = NULL
 * is skipped.  far_amd/optim.py:AdamW builds it once and refreshes the gradient pointers every step. */
long far_adamw_table_bytes(int n, long nblocks);
int far_adamw_step_f32(const void* table, int n, long nblocks, double lr, double beta1, double beta2, double eps, double wd, double bc1,
                       double bc2_sqrt, far_stream_t stream);

/* K19: BatchNorm2d with BATCH statistics (training mode) on channels_last fp32 tensors [M = N H W][C], C % 4 == 0, C <= 1024
 * (batchnorm_train_f32.hip) -- resnet_fpn.py:24-41, 60-62, 75-91 under autograd; deterministic (fixed-order two-stage sums).
 *   far_bn_train_stats_f32: scale[c] = gamma[c] rstd[c], shift[c] = beta[c] - mean[c] scale[c] (the normalisation + activation +
 *     residual is then far_affine_act_f32), mean_out / rstd_out for the backward; running_mean / running_var updated in place as
 *     nn.BatchNorm2d(momentum) does (biased variance for the normalisation, unbiased for the running estimate); NULL gamma / beta = 1 / 0.
 *   far_bn_train_bwd_f32: backward of y = act(bn(x) (+ residual)): g = dy act'(y) (act 0 none, 1 ReLU, 2 LeakyReLU(slope); y read
 *     only when act != 0), dbeta = sum g, dgamma = sum g xhat, dx = gamma rstd (g - dbeta / M - xhat dgamma / M), dres = g when
 *     dres != NULL.  ws: far_bn_train_ws_bytes(M, C) bytes of device scratch. */
long far_bn_train_ws_bytes(long M, int C);
/* far_bn_act_train_fwd_f32: the whole training forward in one call (statistics, then y = act(x scale + shift (+ res))); vec = 4 C
 * floats { scale, shift, mean, rstd } kept for the backward. */
int far_bn_act_train_fwd_f32(const float* x, const float* res, long M, int C, const float* gamma, const float* beta, float eps,
                             float momentum, float* running_mean, float* running_var, int act, float slope, float* y, float* vec,
                             void* ws, long ws_bytes, far_stream_t stream);
int far_bn_train_stats_f32(const float* x, long M, int C, const float* gamma, const float* beta, float eps, float momentum,
                           float* running_mean, float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out,
                           void* ws, long ws_bytes, far_stream_t stream);
int far_bn_train_bwd_f32(const float* x, const float* dy, const float* y, const float* mean, const float* rstd, const float* gamma,
                         long M, int C, int act, float slope, float* dx, float* dgamma, float* dbeta, float* dres, void* ws,
                         long ws_bytes, far_stream_t stream);

/* Gradient of that upsampling with respect to lo: dlo [N][h][w][C] from dout [N][2h][2w][C] (channels_last memory, C % 4 == 0),
 * gathered in a fixed order -- bit-identical from run to run.  Replaces the autograd node torch records for
 * F.interpolate(..., scale_factor=2, mode='bilinear', align_corners=True) at resnet_fpn.py:108,113 (its ATen backward scatters
 * with atomics). */
int far_upsample2x_bwd_f32(const float* dout, int N, int h, int w, int C, float* dlo, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K9  implicit-GEMM convolution / linear layer on the f16 matrix cores with split-precision operands
 * replaces (inference) src/loftr/backbone/resnet_fpn.py:5-12 (conv1x1, conv3x3), :15-43 (BasicBlock.forward),
 *      :101-119 (ResNetFPN_8_2.forward) and the nn.Linear layers of src/loftr/loftr_module/transformer.py:25-35
 * Every fp32 operand is split v = hi + lo (two fp16), the product is hi.hi + hi.lo + lo.hi accumulated in fp32:
 * fp32-grade results at 16/3 of the fp32 matrix rate.  split = 0 keeps hi only (plain fp16 operands).
 * --------------------------------------------------------------------------------------------------- */

/* Bytes of the packed image of a [Cout][Cin][ksize][ksize] weight; ksize 1 or 3, stride 1 (or 2 with ksize 3: the
 * image depends on the tile shape the kernel uses for that stride); 0 on bad arguments. */
size_t far_conv_packed_bytes(int Cin, int Cout, int ksize, int stride, int split);

/* Packs w (torch layout [Cout][Cin][ksize][ksize] fp32), multiplied by 2^w_exp, into `packed`.  Choose w_exp with
 * 2^13 <= max|w| 2^w_exp < 2^15 and fold 2^-(w_exp + 4) into the `scale` vector of far_conv_nhwc_f32. */
int far_conv_pack_f32(const float* w, int Cin, int Cout, int ksize, int stride, int w_exp, int split, void* packed,
                      far_stream_t stream);
/* Device-side pack for training (weights change every step; nothing is read back, nothing is copied):
 *   far_weight_scale_f32: scale_out = { 2^w_exp, 2^-(w_exp + 4) } from max|w| over n contiguous floats (one launch);
 *   far_conv_pack_view_f32: packs the weight read through element strides (s_co, s_ci, s_tap; w = the element of tap 0 in
 *   execution order) times scale_in[0].  Forward image of a contiguous [Cout][Cin][k][k] weight: (Cin k k, k k, 1).  Its dgrad
 *   image (the 'same' stride-1 convolution of the output gradient that gives the input gradient, resnet_fpn.py:5-12 under
 *   autograd): Cin / Cout exchanged, strides (k k, Cout_fwd... see far_amd/ops/packs.py:PackedConv.dgrad_view), taps reversed -- the
 *   same tensor, no flipped / transposed copy.  A Linear layer's transposed weight W^T: (1, K_fwd, 0). */
int far_weight_scale_f32(const float* w, long n, float* scale_out, far_stream_t stream);
int far_conv_pack_view_f32(const float* w, long s_co, long s_ci, long s_tap, int Cin, int Cout, int ksize, int stride, int split,
                           const float* scale_in, void* packed, far_stream_t stream);
/* ... and the launch's epilogue scale vector in the same kernel: scale_vec_out[co] = base_scale[co] (1 when NULL) * scale_in[1],
 * what far_conv_nhwc_f32 takes as `scale` (a training step re-packs every weight after every optimizer update). */
int far_conv_pack_view_scaled_f32(const float* w, long s_co, long s_ci, long s_tap, int Cin, int Cout, int ksize, int stride, int split,
                                  const float* scale_in, void* packed, const float* base_scale, float* scale_vec_out,
                                  far_stream_t stream);
/* The same with w_exp chosen on the device from max|w| (no host read: training re-packs every layer after every optimizer
 * step): scale_out (2 device floats) = { 2^w_exp, 2^-(w_exp + 4) }; the caller multiplies its `scale` vector by scale_out[1]. */
int far_conv_pack_auto_f32(const float* w, int Cin, int Cout, int ksize, int stride, int split, void* packed, float* scale_out,
                           far_stream_t stream);
/* Every weight image of a model in two launches (training: all weights change at every optimizer step).  One far_pack_item per
 * image, with far_conv_pack_view_scaled_f32's meaning; scale_owner = index of the item whose max|w| reduction this image uses (its
 * own index: it reduces w_all[0 .. n_all) into pack_scale; a dgrad / transposed image names its forward image and passes the same
 * pack_scale pointer).  far_pack_table_build writes the device table (far_pack_table_bytes(n) bytes) once; far_pack_table_run
 * re-packs all n images from the weights' current values.  n <= 4096. */
typedef struct far_pack_item {
    const float* w;
    long s_co, s_ci, s_tap;
    int Cin, Cout, ksize, stride, split, scale_owner;
    const float* w_all;
    long n_all;
    float* pack_scale;
    void* packed;
    const float* base_scale;
    float* scale_vec;
} far_pack_item;
long far_pack_table_bytes(int n);
int far_pack_table_build(const far_pack_item* items, int n, void* table_dev, far_stream_t stream);
int far_pack_table_run(const void* table_dev, int n, far_stream_t stream);

/* One K9 launch.  y = LN?( act(scale[co] * conv(X, W)[.., co] + shift[co] (+ res)) ) (+ post_res):
 *   X = x [N][H][W][Cin1] or, with x2 != NULL, the channel concatenation [x | x2] (x2 [N][H][W][Cin - Cin1],
 *   Cin1 % 8 == 0; never materialised: transformer.py:64 torch.cat); Cin1 = Cin when x2 == NULL.
 *   y [N][Ho][Wo][Cout], Ho = (H - 1) / stride + 1; fp32 NHWC, zero padding ksize / 2, Cin % 4 == 0; stride 1, or 2
 *   with ksize 3 (resnet_fpn.py:19 conv3x3(in_planes, planes, stride)); ksize 1 with stride 2 = the 1x1 convolution of
 *   x[:, ::2, ::2] read in place (resnet_fpn.py:26-29, the down-sampling shortcut conv1x1(in_planes, planes, stride=2)): `packed`
 *   is the stride-1 image of the weight; no x2 / up / res_group.  A linear layer y = x W^T + b is ksize = 1,
 *   N = H = 1, W = rows, shift = b.
 *   packed / scale: from far_conv_pack_f32 (scale includes 2^-(w_exp + 4)); shift, res may be NULL.
 *   act: 0 none, 1 ReLU, 2 LeakyReLU(slope).  split: 1 = hi + lo operand pairs, 0 = plain fp16 operands.
 *   out_planes > 1 splits the output channels into that many separate contiguous tensors, y = [out_planes][N][Ho][Wo]
 *   [Cout / out_planes] (fused projections: transformer.py:45-47 q_proj / k_proj / v_proj in one launch).
 *   res: y's layout; or, with res_group = G > 1 (ksize 1), [N H W / G][Cout] with row pix / G added to pixel pix -- one
 *   residual row per group of G consecutive rows (fine_preprocess.py:52-57: repeat(feat_c_win, 'n c -> n ww c') +
 *   Linear, without materialising the repeat or the concatenation).
 *   ln_gamma, ln_beta != NULL (Cout = 128 or 256): LayerNorm over the Cout channels (eps ln_eps, biased variance)
 *   times gamma plus beta, then + post_res if given -- transformer.py:61 (norm1 after merge) and :65-67
 *   (x + norm2(mlp(...))) fused into the Linear layer's epilogue.
 *   up [N][H/2][W/2][Cout] (ksize 1, even H, W >= 32, no other residual): its 2x bilinear upsampling
 *   (align_corners = True) is added before the activation -- the FPN merge of resnet_fpn.py:108-109, :113-114
 *   (F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) + lateral 1x1 convolution) in one launch.
 *   act_exp / overflow: the activation range of the split.  Inputs are multiplied by 2^act_exp before hi = fp16(.),
 *   lo = fp16(. - hi): |input| <= 65504 / 2^act_exp is representable, beyond it hi = inf and every output it feeds is
 *   inf / NaN (resnet_fpn.py's plain fp32 convolutions have no such limit: callers lower act_exp and re-launch when
 *   *overflow comes back set -- far_amd/loftr/model.py does, per forward).  `scale` always folds 2^-4; the kernel applies
 *   the correction 2^(4 - act_exp).  Lower exponents cost resolution at the small end only: values below
 *   2^-14 2^-act_exp lose their lo part, i.e. the absolute error floor is 2^-25 2^-act_exp.
 *   y must alias none of the inputs. */
typedef struct far_conv_desc {
    const float* x;
    const float* x2;
    const void* packed;
    const float* scale;
    const float* shift;
    const float* res;
    const float* ln_gamma;
    const float* ln_beta;
    const float* post_res;
    const float* up;
    float* y;
    long N;
    int H, W, Cin, Cin1, Cout, ksize, stride;
    int act, split, out_planes, res_group;
    float slope, ln_eps;
    int act_exp;      /* activations x 2^act_exp before the fp16 split; 4 = default, [-24, 8]: inputs up to 65504 / 2^act_exp */
    int* overflow;    /* device int, |= 1 when an accumulator of the launch is not finite (input out of that range); or NULL */
    const float* act_scale_dev;   /* NULL, or two device floats { 2^e, 2^(4 - e) } from far_grad_scale_f32 that replace act_exp */
} far_conv_desc;

int far_conv_nhwc_f32(const far_conv_desc* desc, far_stream_t stream);

/* The k and v projections of a LoFTREncoderLayer at d_model 256 / 8 heads (transformer.py:52-56 with linear_attention.py:38-45) in
 * ONE launch that never writes k or v: K9's Linear mode ending in the K'^T V product instead of a store.
 * desc: a Linear layer (ksize 1, N = H = 1, W = rows, split = 1) with Cout = 512, out_planes = 2, no activation / residual /
 * LayerNorm; its packed weight = Wk and Wv interleaved head by head (weight rows 64 j + [0, 32) = Wk's rows of head j,
 * 64 j + [32, 64) = Wv's); y unused (NULL).  rows = n_img * S tokens, image after image, S >= 64 (any S: lengths that are no
 * multiple of 64 run on a padded launch geometry, so that a 64-row partial sum never holds rows of two images).
 * kv [n_img][256][33] = the state far_linear_attention_apply_f32 consumes (what far_linear_attention_f32 builds from k and v in
 * its first two launches); fixed summation order per image (its 64-row blocks in order): run-to-run and batch-size independent bits.
 * ws: far_linear_kv_workspace_bytes(rows, S) bytes. */
size_t far_linear_kv_workspace_bytes(long rows, int S);
size_t far_linear_kv_image_bytes(long n_img);
/* kv_img (optional, far_linear_kv_image_bytes(n_img) bytes): the same state as the MFMA operands (fp16 hi / lo pairs x 2^act_exp) and
 * ksum that far_linear_q_apply_f16s reads. */
int far_linear_kv_f16s(const far_conv_desc* desc, int S, void* ws, float* kv, void* kv_img, far_stream_t stream);

/* The q projection of the same layer (transformer.py:51, 54) with LinearAttention's second half (linear_attention.py:46-50) in its
 * epilogue: desc = a Linear layer (ksize 1, N = H = 1, W = rows, split = 1, Cout = 256, out_planes = 1, no activation / residual /
 * LayerNorm) with the plain Wq image; y [rows][256] receives the attention MESSAGE (Q' KV) Z S -- q is never stored and
 * far_linear_attention_apply_f32's launch disappears.  kv_img: far_linear_kv_f16s's image of the SOURCE tokens under the same
 * act_exp (image i serves rows [i L, (i + 1) L)); L = tokens per image on the query side (>= 64), S = the source's length. */
int far_linear_q_apply_f16s(const far_conv_desc* desc, int L, int S, const void* kv_img, float eps, far_stream_t stream);

/* merge_feat of FinePreprocess (fine_preprocess.py:40-57) without the window tensor: K9's Linear mode reading its input rows
 * straight from the fine feature map.  desc: a Linear layer (ksize 1, N = H = 1, W = rows, split = 1; shift / res / res_group / act
 * as far_conv_nhwc_f32) with desc.x = the fine map [n_img][Hf][Wf][Cin] (NHWC); row r of the launch is token r % (W W) of window
 * r / (W W): pixel (cy stride - W / 2 + ky, cx stride - W / 2 + kx) of image b_ids[window], zero outside the map, (cy, cx) =
 * divmod(cell_ids[window], wc), (ky, kx) = divmod(token, W) -- the rows far_fine_gather_f32 would have written for the launch to
 * read back.  rows % (W W) == 0, W <= 15. */
int far_linear_gather_f16s(const far_conv_desc* desc, const int64_t* b_ids, const int64_t* cell_ids, int wc, int W, int stride,
                           long n_img, int Hf, int Wf, far_stream_t stream);

/* K17: the stride-1 3x3 convolutions as Winograd F(2x2, 3x3) on the f16 matrix cores with split operands (conv_wino_f16s.hip):
 * far_conv_nhwc_f32's contract for ksize = 3, stride = 1, split = 1 at 2.25x fewer matrix instructions -- resnet_fpn.py:5-12
 * (conv3x3), :15-43 (BasicBlock), :101-119 (layer*_outconv2).
 *   far_wino_packed_bytes: bytes of the weight image [64-channel block][16-channel k-step][transform rows {0,1} | {2,3}][32 KiB].
 *   far_wino_pack_view_scaled_f32: packs U = G g G^T (float64, then x scale_in[0] and the fp16 hi / lo split) of a [Cout][Cin][3][3]
 *   weight read through element strides like far_conv_pack_view_scaled_f32 (w = the element of tap 0; contiguous: 9 Cin, 9, 1);
 *   scale_in = { 2^w_exp, 2^-(w_exp + 4) } from far_weight_scale_f32; scale_vec_out[co] = base_scale[co] (1 when NULL) x scale_in[1]
 *   (may be NULL).
 *   far_conv3x3_wino_f32: one launch; desc as far_conv_nhwc_f32 with `packed` a Winograd image, ksize 3, stride 1, split 1, and
 *   x2 / ln_* / post_res / up / act_scale_dev NULL, out_planes = res_group = 1, act_exp >= 0 (else -22).  The activations are
 *   split unscaled (|input| <= 16376: the transformed operand is at most 4 |input|); *overflow is raised beyond that, as by K9. */
size_t far_wino_packed_bytes(int Cin, int Cout);
int far_wino_pack_view_scaled_f32(const float* w, long s_co, long s_ci, long s_tap, int Cin, int Cout, const float* scale_in,
                                  void* packed, const float* base_scale, float* scale_vec_out, far_stream_t stream);
int far_conv3x3_wino_f32(const far_conv_desc* desc, far_stream_t stream);
/* Activation scale for an input of unknown magnitude (the output gradient in a dgrad launch): out2 = { 2^e, 2^(4 - e) } with
 * max|x| 2^e in [2^9, 2^10), computed on the device -- pass out2 as far_conv_desc.act_scale_dev. */
int far_grad_scale_f32(const float* x, long n, float* out2, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K16  weight gradient of the 3x3 / 1x1 convolutions and of the Linear layers (training)
 * replaces autograd's backward-weights of src/loftr/backbone/resnet_fpn.py:5-12 (conv1x1 / conv3x3: stride 1 or 2, 'same'
 *      padding, no bias) and of the nn.Linear layers of src/loftr/loftr_module/transformer.py:25-35
 *   dw [Cout][Cin][k][k] = sum over pixels of dy[n][oy][ox][co] * x[n][s oy + ky - k/2][s ox + kx - k/2][ci];  x [N][H][W][Cin],
 *   dy [N][Ho][Wo][Cout] NHWC fp32 contiguous, Ho = (H - 1) / s + 1; a Linear layer is k = 1 with its rows factored as any
 *   H x W.  Split-fp16 operands (hi + lo, three MFMAs per product: fp32-grade), summed deterministically: pixel ranges write
 *   partial sums to `ws` (far_conv_wgrad_ws_bytes() bytes of device scratch) and a second kernel adds them in a fixed order.
 *   x is multiplied by 2^act_exp before the split (as in far_conv_nhwc_f32), dy by dy_scale_dev[0] (far_grad_scale_f32's two
 *   device floats; NULL: computed by this call).  overflow: device int OR-ed with 1 when a sum is non-finite (an operand
 *   beyond the split's range), or NULL.  dw is overwritten.
 * --------------------------------------------------------------------------------------------------- */
long far_conv_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int ksize, int stride);
int far_conv_wgrad_f16s(const float* x, const float* dy, int N, int H, int W, int Cin, int Cout, int ksize, int stride, int act_exp,
                        const float* dy_scale_dev, void* ws, long ws_bytes, float* dw, int* overflow, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K10  backbone stem: 7x7 stride-2 convolution of a 1-channel image + folded BatchNorm + ReLU (exact-f32 MFMA)
 * replaces src/loftr/backbone/resnet_fpn.py:60-62, :103   x0 = relu(bn1(conv1(x)))
 * img [N][H][W] fp32, w [Cout][7][7] (torch layout), y [N][(H+1)/2][(W+1)/2][Cout] NHWC; Cout = 64 or 128.
 * scale = shift = NULL: the bare convolution (no BatchNorm fold, no ReLU) -- the training forward, where bn1 follows with
 * batch statistics.  far_stem7x7_wgrad_f32: its weight gradient dw [Cout][7][7] (overwritten) = sum over pixels of
 * dy[n][oy][ox][co] * img[n][2 oy + ky - 3][2 ox + kx - 3] (autograd's backward-weights of resnet_fpn.py:60), exact fp32
 * products, deterministic two-stage sum through `ws` (far_stem7x7_wgrad_ws_bytes() bytes of device scratch).
 * --------------------------------------------------------------------------------------------------- */
int far_stem7x7_nhwc_f32(const float* img, const float* w, const float* scale, const float* shift, int N, int H, int W,
                         int Cout, float* y, far_stream_t stream);
long far_stem7x7_wgrad_ws_bytes(int N, int H, int W, int Cout);
int far_stem7x7_wgrad_f32(const float* img, const float* dy, int N, int H, int W, int Cout, void* ws, long ws_bytes, float* dw,
                          far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K11  per-pair arithmetic between the solver and the regression head (one launch each)
 * far_pose_pack_f64 replaces src/loftr/utils/supervision.py:218-233 (spvs_RT: [R | t] with the identity fallback of
 *   :221-224, E with its identity fallback, the count tensors) and src/utils/metrics.py:83-85 (< 5 matches: counts 0).
 *   R, E [B][9], t [B][3] float64, status / num_after / tight / ultra [B] int32 as far_solver_f64 leaves them;
 *   offsets [B+1] int32.  rt_out [B][12] (3x4 row-major), E_out [B][9] float64; before_out [B] int64;
 *   after_out / tight_out / ultra_out [B] int32.
 * far_pose_features_f32 replaces src/loftr/loftr.py:137-171 (preprocess_helper) with src/losses/loftr_loss.py:7-8,
 *   31-39 (pose_mean_6d / pose_std_6d, compute_normalized_6d): rt [B][12] float64 -> preds (pose, cast to fp32 then
 *   normalised) and inv_preds (4x4 inverse and normalisation in float64, then cast), each [B][9 + n] fp32 where the n
 *   present count vectors cnt_k [B] (elem_bytes_k = 4: int32, 8: int64, 0: absent) are appended as count / 500.
 * --------------------------------------------------------------------------------------------------- */
int far_pose_pack_f64(const double* R, const double* t, const double* E, const int* status, const int* num_after,
                      const int* tight, const int* ultra, const int* offsets, int B, double* rt_out, double* E_out,
                      long* before_out, int* after_out, int* tight_out, int* ultra_out, far_stream_t stream);
int far_pose_features_f32(const double* rt, int B, const void* cnt0, int elem_bytes0, const void* cnt1, int elem_bytes1,
                          const void* cnt2, int elem_bytes2, const void* cnt3, int elem_bytes3, float* preds,
                          float* inv_preds, far_stream_t stream);
/* The head's regressed pose as the next solver round's prior (loftr.py:186-192): pose [B][9] fp32 (normalised [t | 6D rotation]),
 * mean / std [9] fp32 (device) -> prior [B][3][4] fp32 = [ rotation_6d_to_matrix(pose[3:9] std + mean) | pose[0:3] std + mean ]. */
int far_prior_from_pose_f32(const float* pose, const float* mean, const float* stdv, int B, float* prior, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K15  the small dense layers of the regression head, row-independent and in exact fp32 (fixed summation order)
 * replaces, for inference, src/loftr/loftr_module/transformer.py:294-295 (F = v~^T (P v~): the 70 x N x 70 torch.bmm after K2),
 *      :423-431 (encoder 35840 -> 512 -> 512, pose_regressor_simple_moe 512 -> 512 -> 9) and :448-458 (moe_predictor
 *      35862 -> 512 -> 512 -> 2 + sigmoid).  A vendor GEMM chooses its kernel by the row count, so a pair's regressed pose used
 *      to depend (1e-7) on how many pairs shared the batch; here row b of any batch is bit-identical to the pair run alone.
 * far_rows_linear_f32: y[b][:] = act(x[b][:] W^T + bias + add[b][:]); W packed by far_rows_linear_pack_f32 from the torch
 *   [N][K] layout; x / y / add rows ldx / ldy / ld_add floats apart; add, bias may be NULL; act 0 none, 1 ReLU, 2 sigmoid,
 *   3 GELU (erf).  ws: far_rows_linear_workspace_bytes(B, N, K).
 * far_emm_contract_f32: F [Z][70][70] = [v | pos]^T T per problem (v addressed like far_emm_pv_f16s's v: problem
 *   z = p * heads + hh at v + hh * head_stride + p * prob_stride, [N][64]; pos [N][6]; T [Z][N][70]).
 * --------------------------------------------------------------------------------------------------- */
size_t far_rows_linear_packed_bytes(int N, int K);
size_t far_rows_linear_workspace_bytes(int B, int N, int K);
int far_rows_linear_pack_f32(const float* w, int N, int K, void* packed, far_stream_t stream);
int far_rows_linear_f32(const float* x, long ldx, const void* packed, const float* bias, const float* add, long ld_add, int B,
                        int K, int N, int act, float* y, long ldy, void* ws, far_stream_t stream);
size_t far_emm_contract_workspace_bytes(int Z);
int far_emm_contract_f32(const float* v, int heads, long head_stride, long prob_stride, const float* pos, const float* T, int Z,
                         int N, float* F, void* ws, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K4  batched essential-matrix solver (hypothesise / verify / decompose / cheirality), float64
 * replaces src/utils/metrics.py:80-174 (estimate_pose), third_party/prior_ransac/ransac.py:340-442
 *      (RANSAC.forward + verify + get_prior_estimate), cv_geometry.py:713-833 (run_8point),
 *      essential.py:99-139 (decompose_essential_matrix), cv::recoverPose (src/utils/cv2_fcns.py:147-319)
 * ------------------------------------------------------------------------------------------------- */
size_t far_solver_workspace_bytes(int B, int Mtot, int H, int P);

/* Minimal solvers (`minimal`).  8: the normalized 8-point of cv_geometry.py:772-833 (RANSAC model_type 'fundamental',
 *   ransac.py:140-145: sample size 8, score floor 8) for pairs with >= 8 correspondences; pairs with 5..7 -- which the
 *   reference accepts (metrics.py:83-85) and only a five-point solver can fit -- get five-point hypotheses.
 *   5: Nister's five-point solver for every pair (cv_geometry.py:861-1043 run_5point_our_kornia, RANSAC model_type 'essential',
 *   ransac.py:146-150: sample size 5, score floor 5): H / 10 samples, up to ten essential matrices each, model 10 s + k =
 *   real root k of sample s (H >= 10).  Unlike the 8-point it is not degenerate on coplanar correspondences.
 *   H counts the MODELS verified per pair in both modes. */

/* B pairs; pair b owns correspondences [offsets[b], offsets[b+1]) of the concatenated arrays.
 *   kpts0, kpts1  [Mtot][2] fp32 pixel coordinates (mkpts0_f / mkpts1_f); offsets [B+1] int32; Mmax = max count
 *   K0, K1        [B][9] float64 row-major intrinsics
 *   inl_th        [B] float64: squared-Sampson inlier threshold in normalised coordinates
 *                 (3e-7 for prior_ransac: metrics.py:117; (thresh/mean f)^2 for the plain RANSAC branch: :94)
 *   many_thr      != 0: also count inliers at inl_th/10 and /100 (ransac.py:284-287)
 *   priorRT       optional [B][12] fp32 (3x4 row-major): enables biased sampling (ransac.py:358-371) and the
 *                 prior score -err^2/prior_lambda over the P-point cloud pcl [P][3] fp32 (ransac.py:203-231,:397)
 *   H, seed       hypotheses per pair and sampling seed; samples_in optional [B][H][8] int32 overrides sampling
 * outputs (all device): R_out [B][9], t_out [B][3], E_out [B][9] float64; mask_out [Mtot] uint8 (inliers that
 *   also pass cheirality, as cv2.recoverPose leaves its in/out mask); status_out [B] (1 = pose valid, 0 = the
 *   reference's `ret is None`); num_after_out, n_tight_out, n_ultra_out, n_cheir_out, best_out [B] int32.
 *   Optional debug outputs (NULL to skip): F_all_out [B][H][9], count_all_out [B][H], score_all_out [B][H],
 *   samples_out [B][H][8] (minimal = 5: samples_in / samples_out are [B][H / 10][5]).  Mtot == 0 (no pair has a correspondence) is legal: kpts0 / kpts1 / mask_out may then be
 *   NULL and every pair reports status 0. */
int far_solver_f64(const float* kpts0, const float* kpts1, const int* offsets, int B, int Mtot, int Mmax,
                   const double* K0, const double* K1, const double* inl_th, int many_thr,
                   const float* priorRT, const float* pcl, int P, double prior_lambda,
                   int H, int minimal, uint32_t seed, const int* samples_in,
                   double* R_out, double* t_out, double* E_out, uint8_t* mask_out, int* status_out,
                   int* num_after_out, int* n_tight_out, int* n_ultra_out, int* n_cheir_out, int* best_out,
                   double* F_all_out, int* count_all_out, double* score_all_out, int* samples_out,
                   void* ws, far_stream_t stream);

/* The function-level API of the solver (host side: far_amd/ransac.py).
 *
 * far_ransac_f64 = RANSAC(...).forward(kp1, kp2) (third_party/prior_ransac/ransac.py:340-442): stages 1-4 of far_solver_f64
 *   (bias weights, hypotheses, verification, selection) WITHOUT recoverPose, on correspondences already in the coordinates the
 *   model is wanted in (the reference passes K-normalised points, metrics.py:124-127).  Arguments as far_solver_f64.
 *   E_out [B][9] (zeros when no model scored above the minimal sample size: best_model_total stays zeros(3, 3), :354);
 *   mask_out [Mtot]: bit 0 = inlier at inl_th, bit 1 = at inl_th / 10, bit 2 = at inl_th / 100 (:284-287);
 *   n_inl_out / n_tight_out / n_ultra_out / best_out [B] int32 (best = -1: none).  ws: far_solver_workspace_bytes(B, Mtot, H, P).
 * far_eightpoint_f64 = run_8point(points1, points2, weights) (cv_geometry.py:772-833): B problems of N >= 8 correspondences,
 *   p1, p2 [B][N][2], w [B][N] or NULL (ones), F_out [B][9] = normalize_transformation(T2^T F_rank2 T1); float64.
 * far_decompose_essential_f64 = decompose_essential_matrix(E) (essential.py:99-139): E [n][9] -> R1, R2 [n][9], t [n][3]
 *   (sign convention of DESIGN.md "K4": the set {R1, R2} x {t, -t} equals the reference's). */
int far_ransac_f64(const float* kp1, const float* kp2, const int* offsets, int B, int Mtot, int Mmax, const double* inl_th,
                   const float* priorRT, const float* pcl, int P, double prior_lambda, int H, int minimal, uint32_t seed,
                   const int* samples_in, double* E_out, uint8_t* mask_out, int* n_inl_out, int* n_tight_out, int* n_ultra_out,
                   int* best_out, void* ws, far_stream_t stream);
int far_eightpoint_f64(const double* p1, const double* p2, const double* w, int B, int N, double* F_out, far_stream_t stream);
int far_decompose_essential_f64(const double* E, long n, double* R1_out, double* R2_out, double* t_out, far_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K12  correlation-volume warp of the Map-free 6DReg aggregator
 * replaces mapfree_6dreg/lib/models/regression/aggregator.py:44-115 (CorrelationVolumeWarping.forward) in the FAR
 * configuration (POSITION_ENCODER, MAX_SCORE_CHANNEL; config/regression/mapfree/rot6d_trans_with_loftr.yaml):
 *   agg [B][2 D + 3][HW] = cat[vol0, vol1 P^T, grid P^T, rowmax P],  P = softmax_j(vol0_i . vol1_j)  (never materialised)
 * vol0, vol1 [B][D][HW] fp32 (the reference's (B, D, H, W) tensors; D must be 32), grid [2][HW] (meshgrid of
 * linspace(-1, 1, H) x linspace(-1, 1, W), :81-84).  Exact fp32 arithmetic on the f32-input matrix core. */
size_t far_corr_volume_warp_workspace_bytes(int B, int HW);
int far_corr_volume_warp_f32(const float* vol0, const float* vol1, const float* grid, int B, int D, int HW, float* agg,
                             void* ws, far_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FAR_HIP_H */
