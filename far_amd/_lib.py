"""ctypes binding of libfar_hip.so (include/far_hip.h).

The product path has no fallback: if the shared library is missing or a kernel call is attempted on a
non-GPU tensor, this module raises.  `load()` itself needs no GPU (the symbol-export test runs on CPU).
"""
import ctypes
import os

from . import flags

_HERE = os.path.dirname(os.path.abspath(__file__))
# FAR_HIP_LIB: another build of the same library (tools/ab_build.py writes lib/libfar_hip_base.so from a git revision, for
# same-box A/B timings); the default is the in-tree build.
LIB_PATH = flags.value('FAR_HIP_LIB') or os.path.join(_HERE, 'lib', 'libfar_hip.so')

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_d = ctypes.c_double
c_sz = ctypes.c_size_t
c_l = ctypes.c_long
c_u32 = ctypes.c_uint32

# name -> (restype, argtypes); must list every function declared in include/far_hip.h
SIGNATURES = {
    'far_abi_version': (c_i, []),
    'far_last_hip_error': (c_i, []),
    'far_set_tuning': (c_i, [c_i, c_i]),
    'far_mfma_probe_f16': (c_i, [c_i, c_i, c_i, c_p, c_p, c_p]),
    'far_dual_softmax_workspace_bytes': (c_sz, [c_i, c_i, c_i]),
    'far_dual_softmax_stats_f32': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p]),
    'far_coarse_match_f32': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f,
                                   c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'far_coarse_match_bf16_workspace_bytes': (c_sz, [c_i, c_i, c_i, c_i]),
    'far_coarse_match_bf16': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f,
                                    c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'far_coarse_match_f16s_workspace_bytes': (c_sz, [c_i, c_i, c_i, c_i]),
    'far_coarse_match_f16s': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f,
                                    c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'far_conf_matrix_f16s': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    'far_coarse_train_workspace_bytes': (c_sz, [c_i, c_i, c_i, c_i]),
    'far_coarse_pos_conf_f16s': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p]),
    'far_coarse_pos_conf_bwd_f16': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    'far_emm_pv_f32': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p]),
    'far_fine_gather_f32': (c_i, [c_p, c_l, c_l, c_l, c_l, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    'far_fine_scatter_f32': (c_i, [c_p, c_l, c_l, c_l, c_l, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    'far_fine_scatter_det_f32': (c_i, [c_p, c_l, c_l, c_l, c_l, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'far_fine_expect_f32': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_f, c_p, c_p, c_p, c_p, c_p]),
    'far_linear_attention_workspace_bytes': (c_sz, [c_i, c_i, c_i, c_i]),
    'far_linear_attention_f32': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_p, c_p, c_p]),
    'far_linear_attention_apply_f32': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_f, c_p, c_p]),
    'far_linear_attention_bwd_workspace_bytes': (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    'far_linear_attention_bwd_f32': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p]),
    'far_layernorm_f32': (c_i, [c_p, c_p, c_p, c_p, c_l, c_i, c_f, c_p, c_p]),
    'far_layernorm_bwd_ws_bytes': (c_l, [c_l, c_i]),
    'far_layernorm_bwd_f32': (c_i, [c_p, c_p, c_p, c_l, c_i, c_f, c_p, c_p, c_p, c_p, c_l, c_p]),
    'far_attn_block_packed_bytes': (c_sz, [c_i]),
    'far_attn_block_f16s': (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_f, c_p, c_p, c_p]),
    'far_mlp_fused_packed_bytes': (c_sz, [c_i]),
    'far_mlp_fused_f16s': (c_i, [c_p, c_p, c_p, c_l, c_i, c_f, c_f, c_p, c_p, c_f, c_p, c_p, c_p]),
    'far_mlp_fused_f16': (c_i, [c_p, c_p, c_p, c_l, c_i, c_f, c_f, c_p, c_p, c_f, c_p, c_p, c_p]),
    'far_attn_block_f16': (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_f, c_p, c_p, c_p]),
    'far_affine_act_f32': (c_i, [c_p, c_p, c_p, c_p, c_l, c_i, c_l, c_i, c_i, c_f, c_p, c_p]),
    'far_upsample2x_add_f32': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'far_upsample2x_bwd_f32': (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    'far_adamw_table_bytes': (c_l, [c_i, c_l]),
    'far_adamw_step_f32': (c_i, [c_p, c_i, c_l] + [ctypes.c_double] * 7 + [c_p]),
    'far_prior_from_pose_f32': (c_i, [c_p, c_p, c_p, c_i, c_p, c_p]),
    'far_bn_train_ws_bytes': (c_l, [c_l, c_i]),
    'far_bn_act_train_fwd_f32': (c_i, [c_p, c_p, c_l, c_i, c_p, c_p, c_f, c_f, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_l, c_p]),
    'far_bn_train_stats_f32': (c_i, [c_p, c_l, c_i, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p]),
    'far_bn_train_bwd_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_l, c_p]),
    'far_conv_packed_bytes': (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    'far_conv_pack_f32': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'far_weight_scale_f32': (c_i, [c_p, c_l, c_p, c_p]),
    'far_conv_pack_view_f32': (c_i, [c_p, c_l, c_l, c_l, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    'far_enc_layer_saved_floats': (c_l, [c_p]),
    'far_enc_layer_grads_floats': (c_l, [c_p]),
    'far_enc_layer_fwd_ws_bytes': (c_l, [c_p]),
    'far_enc_layer_bwd_ws_bytes': (c_l, [c_p]),
    'far_enc_layer_grads_offsets': (c_i, [c_p, c_p]),
    'far_enc_layer_fwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p]),
    'far_enc_layer_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p]),
    'far_stream_fork': (c_p, [c_p, c_i]),
    'far_stream_join': (c_i, [c_p, c_i]),
    'far_pack_table_bytes': (c_l, [c_i]),
    'far_pack_table_build': (c_i, [c_p, c_i, c_p, c_p]),
    'far_pack_table_run': (c_i, [c_p, c_i, c_p]),
    'far_conv_pack_view_scaled_f32': (c_i, [c_p, c_l, c_l, c_l, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    'far_conv_pack_auto_f32': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    'far_grad_scale_f32': (c_i, [c_p, c_l, c_p, c_p]),
    'far_conv_nhwc_f32': (c_i, [c_p, c_p]),                  # (const far_conv_desc*, stream): see ConvDesc
    'far_linear_kv_workspace_bytes': (c_sz, [c_l, c_i]),
    'far_linear_kv_image_bytes': (c_sz, [c_l]),
    'far_linear_kv_f16s': (c_i, [c_p, c_i, c_p, c_p, c_p, c_p]),
    'far_linear_q_apply_f16s': (c_i, [c_p, c_i, c_i, c_p, c_f, c_p]),
    'far_linear_gather_f16s': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_l, c_i, c_i, c_p]),
    'far_wino_packed_bytes': (c_sz, [c_i, c_i]),
    'far_wino_pack_view_scaled_f32': (c_i, [c_p, c_l, c_l, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    'far_conv3x3_wino_f32': (c_i, [c_p, c_p]),                # (const far_conv_desc*, stream): K17
    'far_emm_pv_f16s_workspace_bytes': (c_sz, [c_i, c_i]),
    'far_emm_pv_f16s': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_l, c_l, c_i, c_p, c_p, c_p, c_p]),
    'far_emm_pv_f16': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_l, c_l, c_i, c_p, c_p, c_p, c_p]),
    'far_emm_pv_f16s_copy_stats': (c_i, [c_p, c_i, c_i, c_p, c_p, c_p]),
    'far_emm_bwd_workspace_bytes': (c_sz, [c_i, c_i]),
    'far_emm_bwd_f16': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p]),
    'far_corr_volume_warp_workspace_bytes': (c_sz, [c_i, c_i]),
    'far_corr_volume_warp_f32': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    'far_conv_wgrad_ws_bytes': (c_l, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    'far_conv_wgrad_f16s': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_p, c_p, c_p]),
    'far_stem7x7_nhwc_f32': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    'far_stem7x7_wgrad_ws_bytes': (c_l, [c_i, c_i, c_i, c_i]),
    'far_stem7x7_wgrad_f32': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_p]),
    'far_pose_pack_f64': (c_i, [c_p] * 8 + [c_i] + [c_p] * 6 + [c_p]),
    'far_pose_features_f32': (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p]),
    'far_rows_linear_packed_bytes': (c_sz, [c_i, c_i]),
    'far_rows_linear_workspace_bytes': (c_sz, [c_i, c_i, c_i]),
    'far_rows_linear_pack_f32': (c_i, [c_p, c_i, c_i, c_p, c_p]),
    'far_rows_linear_f32': (c_i, [c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_p]),
    'far_emm_contract_workspace_bytes': (c_sz, [c_i]),
    'far_emm_contract_f32': (c_i, [c_p, c_i, c_l, c_l, c_p, c_p, c_i, c_i, c_p, c_p, c_p]),
    'far_solver_workspace_bytes': (c_sz, [c_i, c_i, c_i, c_i]),
    'far_solver_f64': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_d, c_i, c_i, c_u32, c_p]
                       + [c_p] * 14 + [c_p, c_p]),
    'far_ransac_f64': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_d, c_i, c_i, c_u32, c_p] + [c_p] * 6 + [c_p, c_p]),
    'far_eightpoint_f64': (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    'far_decompose_essential_f64': (c_i, [c_p, c_l, c_p, c_p, c_p, c_p]),
    'far_build_id': (ctypes.c_char_p, []),
}



class ConvDesc(ctypes.Structure):
    """far_conv_desc of include/far_hip.h (field order and types must match)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ('x', 'x2', 'packed', 'scale', 'shift', 'res', 'ln_gamma', 'ln_beta',
                                               'post_res', 'up', 'y')] + \
               [('N', ctypes.c_long)] + \
               [(n, ctypes.c_int) for n in ('H', 'W', 'Cin', 'Cin1', 'Cout', 'ksize', 'stride', 'act', 'split',
                                            'out_planes', 'res_group')] + \
               [('slope', ctypes.c_float), ('ln_eps', ctypes.c_float), ('act_exp', ctypes.c_int), ('overflow', ctypes.c_void_p),
                ('act_scale_dev', ctypes.c_void_p)]


class PackItem(ctypes.Structure):
    """far_pack_item of include/far_hip.h (field order and types must match)."""
    _fields_ = [('w', ctypes.c_void_p), ('s_co', ctypes.c_long), ('s_ci', ctypes.c_long), ('s_tap', ctypes.c_long)] + \
               [(n, ctypes.c_int) for n in ('Cin', 'Cout', 'ksize', 'stride', 'split', 'scale_owner')] + \
               [('w_all', ctypes.c_void_p), ('n_all', ctypes.c_long), ('pack_scale', ctypes.c_void_p), ('packed', ctypes.c_void_p),
                ('base_scale', ctypes.c_void_p), ('scale_vec', ctypes.c_void_p)]


class EncLayer(ctypes.Structure):
    """far_enc_layer of include/far_hip.h (field order and types must match)."""
    _fields_ = [('bs', ctypes.c_long), ('L', ctypes.c_long), ('S', ctypes.c_long)] + \
               [(n, ctypes.c_int) for n in ('C', 'nhead', 'self_attn', 'split', 'act_exp', 'overlap')] + \
               [(n, ctypes.c_float) for n in ('eps1', 'eps2', 'attn_eps')] + \
               [('img', ctypes.c_void_p * 6), ('img_scale', ctypes.c_void_p * 6), ('imgT', ctypes.c_void_p * 6), ('imgT_scale', ctypes.c_void_p * 6)] + \
               [(n, ctypes.c_void_p) for n in ('g1', 'b1', 'g2', 'b2', 'overflow')]


EXPECTED_ABI = 7          # far_abi_version() of the library these signatures describe (include/far_hip.h)
_lib = None


class FarHipError(RuntimeError):
    pass


def load():
    """Load libfar_hip.so and attach signatures.  Raises FarHipError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FarHipError(
            f'{LIB_PATH} not found: build it with `python -m far_amd.build` (hipcc --offload-arch=gfx950). '
            'far_amd has no CPU or eager fallback for its kernels.')
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.  The HIP runtime that owns torch's streams and
    # allocations must be the one this library binds to, so torch is imported (and its runtime mapped) BEFORE
    # the dlopen: a libfar_hip.so loaded first would pull in /opt/rocm's copy and later launches on torch's
    # streams fail with hipErrorNoDevice (observed; see INTEGRATION.md "load order").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    lib.far_abi_version.restype = ctypes.c_int
    lib.far_abi_version.argtypes = []
    abi = lib.far_abi_version()
    if abi != EXPECTED_ABI:
        # a stale build (or a library of another revision selected through FAR_HIP_LIB) would be called with shifted arguments:
        # silent memory corruption or a wrong stream, not an error
        raise FarHipError(f'{LIB_PATH} has ABI version {abi}, these bindings expect {EXPECTED_ABI}: rebuild it with '
                          '`python -m far_amd.build --force` (or point FAR_HIP_LIB at a library of this revision)')
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    # provenance: the library travels outside git (built in-tree, snapshot-copied to the GPU box).  Its build id must equal the id of
    # the sources it is loaded next to; FAR_HIP_LIB (tools/ab_build.py: a library of another revision, on purpose) skips the check
    if not flags.value('FAR_HIP_LIB'):
        from . import build as _build
        have = lib.far_build_id().decode()
        want = _build.source_id() if os.path.isdir(_build.CSRC) else have          # a deployment without the sources: nothing to compare with
        if have != want:
            raise FarHipError(f'{LIB_PATH} was built from other sources (build id {have}, far_amd/csrc is {want}): rebuild it with '
                              '`python -m far_amd.build`')
    for kv in filter(None, (flags.value('FAR_TUNING') or '').split(',')):      # A/B aid: FAR_TUNING="10=1,8=1" -> far_set_tuning(key, value)
        k, v = kv.split('=')
        lib.far_set_tuning(int(k), int(v))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        detail = f' (hipError_t {_lib.far_last_hip_error()})' if (rc == -5 and _lib is not None) else ''
        raise FarHipError(f'{what} failed with code {rc}{detail}')
