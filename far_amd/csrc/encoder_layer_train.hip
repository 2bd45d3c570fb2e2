// Native driver of one LoFTR encoder layer under training: the whole forward and the whole backward of
// mp3d_loftr/src/loftr/loftr_module/transformer.py:44-67 (LoFTREncoderLayer.forward under autograd) as ONE library call each.
//
// No new arithmetic: the calls below are the library's own entry points (K9 far_conv_nhwc_f32, K5 far_linear_attention_*,
// K6 far_layernorm_*, K16 far_conv_wgrad_f16s, far_grad_scale_f32) in the order far_amd/loftr/layer_train.py issues them from
// Python.  What this buys is the host: at batch 1 a layer's backward is ~36 launches of 5-25 us, and issuing them through
// Python costs ~13 us each (ctypes marshalling, tensor allocation, descriptor construction) -- ~480 us per call against ~450 us
// of kernels, so the step is bound by the interpreter.  From here a launch costs ~2 us, and the layer's
// independent launches overlap on the library's side streams (k / v projections next to q's; every weight gradient next to the
// chain of input gradients).
// Buffers: the caller passes one `saved` buffer (forward -> backward), one `grads` buffer and one scratch `ws`, laid out by
// the *_floats / *_bytes queries below; nothing is allocated here.
#include "common.h"

extern "C" {
struct far_conv_desc {          // mirrors include/far_hip.h
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
    int act_exp;
    int* overflow;
    const float* act_scale_dev;
};
int far_conv_nhwc_f32(const far_conv_desc* desc, hipStream_t stream);
size_t far_linear_attention_workspace_bytes(int N, int S, int H, int D);
int far_linear_attention_f32(const float* q, const float* k, const float* v, int N, int L, int S, int H, int D, const uint8_t* q_mask,
                             const uint8_t* kv_mask, float eps, float* out, void* ws, hipStream_t stream);
size_t far_linear_attention_bwd_workspace_bytes(int N, int L, int S, int H, int D);
int far_linear_attention_bwd_f32(const float* q, const float* k, const float* v, const float* g, int N, int L, int S, int H, int D,
                                 const uint8_t* q_mask, const uint8_t* kv_mask, float eps, float* dq, float* dk, float* dv, void* ws,
                                 hipStream_t stream);
int far_layernorm_f32(const float* x, const float* gamma, const float* beta, const float* res, long rows, int C, float eps, float* y,
                      hipStream_t stream);
long far_layernorm_bwd_ws_bytes(long rows, int C);
int far_layernorm_bwd_f32(const float* x, const float* gamma, const float* dy, long rows, int C, float eps, float* dx, float* dgamma,
                          float* dbeta, void* ws, long ws_bytes, hipStream_t stream);
int far_grad_scale_f32(const float* x, long n, float* out2, hipStream_t stream);
long far_conv_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int ksize, int stride);
int far_conv_wgrad_f16s(const float* x, const float* dy, int N, int H, int W, int Cin, int Cout, int ksize, int stride, int act_exp,
                        const float* dy_scale_dev, void* ws, long ws_bytes, float* dw, int* overflow, hipStream_t stream);
void* far_stream_fork(hipStream_t main, int i);
int far_stream_join(hipStream_t main, int i);

// One encoder layer: x (bs, L, C), source (bs, S, C) (self_attn: source = x, S = L); weight images in the order
// q, k, v, merge, mlp[0], mlp[2]: img[i] / img_scale[i] = the forward image and its epilogue scale vector (PackedConv.packed /
// .scale), imgT / imgT_scale = the transposed (dgrad) image.
struct far_enc_layer {          // mirrors include/far_hip.h
    long bs, L, S;
    int C, nhead, self_attn, split, act_exp, overlap;
    float eps1, eps2, attn_eps;
    const void* img[6];
    const float* img_scale[6];
    const void* imgT[6];
    const float* imgT_scale[6];
    const float *g1, *b1, *g2, *b2;
    int* overflow;
};
}  // extern "C"

namespace {

inline long al64(long n) { return (n + 63) & ~63L; }          // floats: 256-byte alignment of every piece

struct Saved {                  // float offsets into `saved`
    long q, k, v, msg0, m1, xcat, h, m2, total;
};
Saved saved_layout(const far_enc_layer& d) {
    const long R = d.bs * d.L, Rs = d.bs * d.S, C = d.C;
    Saved s;
    long o = 0;
    auto take = [&](long n) { const long r = o; o += al64(n); return r; };
    s.q = take(R * C); s.k = take(Rs * C); s.v = take(Rs * C); s.msg0 = take(R * C); s.m1 = take(R * C);
    s.xcat = take(R * 2 * C); s.h = take(R * 2 * C); s.m2 = take(R * C);
    s.total = o;
    return s;
}

struct Grads {                  // float offsets into `grads`
    long dx, ds, dw[6], dg1, db1, dg2, db2, total;
};
Grads grads_layout(const far_enc_layer& d) {
    const long R = d.bs * d.L, Rs = d.bs * d.S, C = d.C;
    Grads g;
    long o = 0;
    auto take = [&](long n) { const long r = o; o += al64(n); return r; };
    g.dx = take(R * C);
    g.ds = d.self_attn ? -1 : take(Rs * C);
    for (int i = 0; i < 4; ++i) g.dw[i] = take(C * C);
    g.dw[4] = take(4 * C * C);
    g.dw[5] = take(2 * C * C);
    g.dg1 = take(C); g.db1 = take(C); g.dg2 = take(C); g.db2 = take(C);
    g.total = o;
    return g;
}

inline long wgrad_ws(long rows, int K, int N) {               // as far_amd/ops/conv.py:linear_wgrad factors the rows
    const int h = rows % 32 == 0 ? (int)(rows / 32) : 1;
    return far_conv_wgrad_ws_bytes(1, h, (int)(rows / h), K, N, 1, 1);
}

struct Scratch {                // byte offsets into `ws` of the backward
    long dm2, dh, dcat, dxa, dm1, dmsg, dq, dk, dv, t1, t2, scales, ln, wg, att, total;
    long ln_bytes, wg_bytes;
};
Scratch scratch_layout(const far_enc_layer& d) {
    const long R = d.bs * d.L, Rs = d.bs * d.S, C = d.C;
    Scratch s;
    long o = 0;
    auto take = [&](long bytes) { const long r = o; o += (bytes + 255) & ~255L; return r; };
    s.dm2 = take(R * C * 4); s.dh = take(R * 2 * C * 4); s.dcat = take(2 * R * C * 4); s.dxa = take(R * C * 4);
    s.dm1 = take(R * C * 4); s.dmsg = take(R * C * 4); s.dq = take(R * C * 4); s.dk = take(Rs * C * 4); s.dv = take(Rs * C * 4);
    s.t1 = take((R > Rs ? R : Rs) * C * 4); s.t2 = take((R > Rs ? R : Rs) * C * 4);
    s.scales = take(6 * 256);
    s.ln_bytes = far_layernorm_bwd_ws_bytes(R, d.C);
    s.ln = take(s.ln_bytes);
    long wg = 0;
    const long cand[4] = {wgrad_ws(R, 2 * d.C, d.C), wgrad_ws(R, 2 * d.C, 2 * d.C), wgrad_ws(R, d.C, d.C), wgrad_ws(Rs, d.C, d.C)};
    for (long c : cand) wg = c > wg ? c : wg;
    s.wg_bytes = wg;
    s.wg = take(6 * wg);                                      // one per weight gradient: they run on a side stream, out of step with `ws` users
    s.att = take((long)far_linear_attention_bwd_workspace_bytes((int)d.bs, (int)d.L, (int)d.S, d.nhead, d.C / d.nhead));
    s.total = o;
    return s;
}

bool bad_layer(const far_enc_layer* d) {
    if (!d || d->bs <= 0 || d->L <= 0 || d->S <= 0 || d->C <= 0 || (d->C & 3) || d->C > 512 || d->nhead <= 0 || d->C % d->nhead) return true;
    const int D = d->C / d->nhead;
    if (D != 16 && D != 32) return true;
    if (d->self_attn && d->S != d->L) return true;
    if (d->bs * (d->L > d->S ? d->L : d->S) > 0x7fffffffL / 4) return true;
    for (int i = 0; i < 6; ++i)
        if (!d->img[i] || !d->img_scale[i] || !d->imgT[i] || !d->imgT_scale[i]) return true;
    return !d->g1 || !d->b1 || !d->g2 || !d->b2;
}

// y (rows, Cout) = act(x (rows, Cin) W^T * scale (+ res)): K9 as a Linear layer
int linear(const far_enc_layer& d, const float* x, long rows, int Cin, int Cout, const void* img, const float* scale, float* y, int act,
           const float* res, int out_planes, const float* act_scale_dev, hipStream_t st) {
    far_conv_desc c = {};
    c.x = x; c.packed = img; c.scale = scale; c.res = res; c.y = y;
    c.N = 1; c.H = 1; c.W = (int)rows; c.Cin = Cin; c.Cin1 = Cin; c.Cout = Cout; c.ksize = 1; c.stride = 1;
    c.act = act; c.split = d.split; c.out_planes = out_planes; c.res_group = 1; c.slope = 0.01f; c.ln_eps = 0.f;
    c.act_exp = d.act_exp; c.overflow = d.overflow; c.act_scale_dev = act_scale_dev;
    return far_conv_nhwc_f32(&c, st);
}

__global__ void k_cat2(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, long rows, int c4) {
    const long total = rows * 2 * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / (2 * c4);
        const int c = (int)(i - r * 2 * c4);
        out[i] = c < c4 ? a[r * c4 + c] : b[r * c4 + c - c4];
    }
}
// dh = h > 0 ? dh : 0  (aten::threshold_backward(dh, h, 0))
__global__ void k_relu_bwd(float4* __restrict__ dh, const float4* __restrict__ h, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 g = dh[i];
        const float4 v = h[i];
        g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f; g.z = v.z > 0.f ? g.z : 0.f; g.w = v.w > 0.f ? g.w : 0.f;
        dh[i] = g;
    }
}
__global__ void k_add(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 u = a[i], v = b[i];
        out[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}
inline unsigned ew_blocks(long n4) {
    long b = (n4 + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

#define FAR_TRY(call) do { const int rc_ = (call); if (rc_ != FAR_OK) return rc_; } while (0)

// Joins the forked side streams when its scope ends -- on the error returns of FAR_TRY too: the caller frees `ws`, `saved` and
// `grads` as soon as the call returns, and work forked onto a side stream must not still be reading them then.
struct SideJoin {
    hipStream_t main;
    unsigned forked = 0;               // bit i: side stream i holds work since the last join
    explicit SideJoin(hipStream_t m) : main(m) {}
    void* fork(int i) { forked |= 1u << i; return far_stream_fork(main, i); }
    int join(int i) { forked &= ~(1u << i); return far_stream_join(main, i); }
    ~SideJoin() { for (int i = 0; i < 4; ++i) if (forked & (1u << i)) (void)far_stream_join(main, i); }
};

}  // namespace

extern "C" {

long far_enc_layer_saved_floats(const far_enc_layer* d) { return bad_layer(d) ? 0 : saved_layout(*d).total; }
long far_enc_layer_grads_floats(const far_enc_layer* d) { return bad_layer(d) ? 0 : grads_layout(*d).total; }
long far_enc_layer_fwd_ws_bytes(const far_enc_layer* d) {
    return bad_layer(d) ? 0 : (long)far_linear_attention_workspace_bytes((int)d->bs, (int)d->S, d->nhead, d->C / d->nhead) + 256;
}
long far_enc_layer_bwd_ws_bytes(const far_enc_layer* d) { return bad_layer(d) ? 0 : scratch_layout(*d).total; }
// Float offsets of the pieces of `grads`: out[0..11] = dx, ds (-1 for self-attention), dW q, k, v, merge, mlp0, mlp2, dgamma1,
// dbeta1, dgamma2, dbeta2.
int far_enc_layer_grads_offsets(const far_enc_layer* d, long* out12) {
    if (bad_layer(d) || !out12) return FAR_EINVAL;
    const Grads g = grads_layout(*d);
    out12[0] = g.dx; out12[1] = g.ds;
    for (int i = 0; i < 6; ++i) out12[2 + i] = g.dw[i];
    out12[8] = g.dg1; out12[9] = g.db1; out12[10] = g.dg2; out12[11] = g.db2;
    return FAR_OK;
}

// y (bs, L, C) = LoFTREncoderLayer(x, source); `saved` (far_enc_layer_saved_floats) keeps what the backward needs.
int far_enc_layer_fwd(const far_enc_layer* d, const float* x, const float* source, float* saved, float* y, void* ws, long ws_bytes,
                      hipStream_t stream) {
    far_clear_errors();
    if (bad_layer(d) || !x || !saved || !y || !ws || ws_bytes < far_enc_layer_fwd_ws_bytes(d) || (!d->self_attn && !source)) return FAR_EINVAL;
    const far_enc_layer& L = *d;
    const float* src = L.self_attn ? x : source;
    const long R = L.bs * L.L, Rs = L.bs * L.S;
    const int C = L.C;
    const Saved s = saved_layout(L);
    float *q = saved + s.q, *k = saved + s.k, *v = saved + s.v, *msg0 = saved + s.msg0, *m1 = saved + s.m1, *xcat = saved + s.xcat,
          *h = saved + s.h, *m2 = saved + s.m2;
    // q | k | v: independent, each a fraction of the CUs -> k and v on side streams
    hipStream_t sk = stream, sv = stream;
    SideJoin sides(stream);
    if (L.overlap) {
        sk = (hipStream_t)sides.fork(0);
        sv = (hipStream_t)sides.fork(1);
        if (!sk || !sv) return FAR_ELAUNCH;
    }
    FAR_TRY(linear(L, src, Rs, C, C, L.img[1], L.img_scale[1], k, 0, nullptr, 1, nullptr, sk));
    FAR_TRY(linear(L, src, Rs, C, C, L.img[2], L.img_scale[2], v, 0, nullptr, 1, nullptr, sv));
    FAR_TRY(linear(L, x, R, C, C, L.img[0], L.img_scale[0], q, 0, nullptr, 1, nullptr, stream));
    if (L.overlap) { FAR_TRY(sides.join(0)); FAR_TRY(sides.join(1)); }
    FAR_TRY(far_linear_attention_f32(q, k, v, (int)L.bs, (int)L.L, (int)L.S, L.nhead, C / L.nhead, nullptr, nullptr, L.attn_eps, msg0, ws, stream));
    FAR_TRY(linear(L, msg0, R, C, C, L.img[3], L.img_scale[3], m1, 0, nullptr, 1, nullptr, stream));
    // norm1 into the scratch half of y's buffer would alias: y is free until the end -> use it for norm1's output
    FAR_TRY(far_layernorm_f32(m1, L.g1, L.b1, nullptr, R, C, L.eps1, y, stream));
    hipLaunchKernelGGL(k_cat2, dim3(ew_blocks(R * 2 * (C / 4))), dim3(256), 0, stream, (const float4*)x, (const float4*)y, (float4*)xcat, R, C / 4);
    FAR_TRY(linear(L, xcat, R, 2 * C, 2 * C, L.img[4], L.img_scale[4], h, 1, nullptr, 1, nullptr, stream));
    FAR_TRY(linear(L, h, R, 2 * C, C, L.img[5], L.img_scale[5], m2, 0, nullptr, 1, nullptr, stream));
    FAR_TRY(far_layernorm_f32(m2, L.g2, L.b2, x, R, C, L.eps2, y, stream));
    return far_check_launch();
}

// Every gradient of the layer into `grads` (far_enc_layer_grads_floats / _offsets) from the output gradient gy (bs, L, C).
int far_enc_layer_bwd(const far_enc_layer* d, const float* x, const float* source, const float* saved, const float* gy, float* grads,
                      void* ws, long ws_bytes, hipStream_t stream) {
    far_clear_errors();
    if (bad_layer(d) || !x || !saved || !gy || !grads || !ws || ws_bytes < far_enc_layer_bwd_ws_bytes(d) || (!d->self_attn && !source))
        return FAR_EINVAL;
    const far_enc_layer& L = *d;
    const float* src = L.self_attn ? x : source;
    const long R = L.bs * L.L, Rs = L.bs * L.S;
    const int C = L.C;
    const Saved s = saved_layout(L);
    const Grads g = grads_layout(L);
    const Scratch w = scratch_layout(L);
    const float *q = saved + s.q, *k = saved + s.k, *v = saved + s.v, *msg0 = saved + s.msg0, *m1 = saved + s.m1, *xcat = saved + s.xcat,
                *h = saved + s.h, *m2 = saved + s.m2;
    char* const wb = reinterpret_cast<char*>(ws);
    auto F = [&](long off) { return reinterpret_cast<float*>(wb + off); };
    float *dm2 = F(w.dm2), *dh = F(w.dh), *dcat = F(w.dcat), *dxa = F(w.dxa), *dm1 = F(w.dm1), *dmsg = F(w.dmsg), *dq = F(w.dq),
          *dk = F(w.dk), *dv = F(w.dv), *t1 = F(w.t1), *t2 = F(w.t2);
    auto scale = [&](int i) { return F(w.scales + 256 * i); };
    int nw = 0;
    SideJoin sides(stream);
    // a weight gradient: on side stream 0 (behind everything `stream` holds now), its own scratch slab
    auto wgrad = [&](const float* xin, const float* dy, long rows, int K, int N, const float* sc, float* dw) -> int {
        hipStream_t st = stream;
        if (L.overlap) {
            st = (hipStream_t)sides.fork(0);
            if (!st) return FAR_ELAUNCH;
        }
        const int hh = rows % 32 == 0 ? (int)(rows / 32) : 1;
        return far_conv_wgrad_f16s(xin, dy, 1, hh, (int)(rows / hh), K, N, 1, 1, L.act_exp, sc, wb + w.wg + w.wg_bytes * (nw++), w.wg_bytes, dw,
                                   L.overflow, st);
    };
    // norm2 (+ residual: its gradient is gy itself), mlp[2]
    FAR_TRY(far_layernorm_bwd_f32(m2, L.g2, gy, R, C, L.eps2, dm2, grads + g.dg2, grads + g.db2, wb + w.ln, w.ln_bytes, stream));
    FAR_TRY(far_grad_scale_f32(dm2, R * C, scale(0), stream));
    FAR_TRY(wgrad(h, dm2, R, 2 * C, C, scale(0), grads + g.dw[5]));
    FAR_TRY(linear(L, dm2, R, C, 2 * C, L.imgT[5], L.imgT_scale[5], dh, 0, nullptr, 1, scale(0), stream));
    // ReLU, mlp[0]: the gradient of cat([x, norm1(..)]) as two planes
    hipLaunchKernelGGL(k_relu_bwd, dim3(ew_blocks(R * 2 * C / 4)), dim3(256), 0, stream, (float4*)dh, (const float4*)h, R * 2 * C / 4);
    FAR_TRY(far_grad_scale_f32(dh, R * 2 * C, scale(1), stream));
    FAR_TRY(wgrad(xcat, dh, R, 2 * C, 2 * C, scale(1), grads + g.dw[4]));
    FAR_TRY(linear(L, dh, R, 2 * C, 2 * C, L.imgT[4], L.imgT_scale[4], dcat, 0, nullptr, 2, scale(1), stream));
    hipLaunchKernelGGL(k_add, dim3(ew_blocks(R * C / 4)), dim3(256), 0, stream, (const float4*)gy, (const float4*)dcat, (float4*)dxa, R * C / 4);
    // norm1, merge
    FAR_TRY(far_layernorm_bwd_f32(m1, L.g1, dcat + R * C, R, C, L.eps1, dm1, grads + g.dg1, grads + g.db1, wb + w.ln, w.ln_bytes, stream));
    FAR_TRY(far_grad_scale_f32(dm1, R * C, scale(2), stream));
    FAR_TRY(wgrad(msg0, dm1, R, C, C, scale(2), grads + g.dw[3]));
    FAR_TRY(linear(L, dm1, R, C, C, L.imgT[3], L.imgT_scale[3], dmsg, 0, nullptr, 1, scale(2), stream));
    // attention core
    FAR_TRY(far_linear_attention_bwd_f32(q, k, v, dmsg, (int)L.bs, (int)L.L, (int)L.S, L.nhead, C / L.nhead, nullptr, nullptr, L.attn_eps, dq, dk, dv,
                                         wb + w.att, stream));
    // the three projections: input gradients accumulate through the dgrad launches' residual input
    FAR_TRY(far_grad_scale_f32(dq, R * C, scale(3), stream));
    FAR_TRY(far_grad_scale_f32(dk, Rs * C, scale(4), stream));
    FAR_TRY(far_grad_scale_f32(dv, Rs * C, scale(5), stream));
    FAR_TRY(wgrad(x, dq, R, C, C, scale(3), grads + g.dw[0]));
    FAR_TRY(wgrad(src, dk, Rs, C, C, scale(4), grads + g.dw[1]));
    FAR_TRY(wgrad(src, dv, Rs, C, C, scale(5), grads + g.dw[2]));
    if (L.self_attn) {
        FAR_TRY(linear(L, dq, R, C, C, L.imgT[0], L.imgT_scale[0], t1, 0, dxa, 1, scale(3), stream));
        FAR_TRY(linear(L, dk, R, C, C, L.imgT[1], L.imgT_scale[1], t2, 0, t1, 1, scale(4), stream));
        FAR_TRY(linear(L, dv, R, C, C, L.imgT[2], L.imgT_scale[2], grads + g.dx, 0, t2, 1, scale(5), stream));
    } else {
        FAR_TRY(linear(L, dq, R, C, C, L.imgT[0], L.imgT_scale[0], grads + g.dx, 0, dxa, 1, scale(3), stream));
        FAR_TRY(linear(L, dk, Rs, C, C, L.imgT[1], L.imgT_scale[1], t1, 0, nullptr, 1, scale(4), stream));
        FAR_TRY(linear(L, dv, Rs, C, C, L.imgT[2], L.imgT_scale[2], grads + g.ds, 0, t1, 1, scale(5), stream));
    }
    if (L.overlap) FAR_TRY(sides.join(0));
    return far_check_launch();
}

}  // extern "C"
