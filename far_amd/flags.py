"""The one registry of far_amd's switches: every FAR_* environment variable and every far_set_tuning key, with what it selects
and what it may change.  Nothing else in the package reads os.environ for a FAR_* name (tests/test_flags.py greps for it), and
tests/test_flags_gpu.py runs every switch and key on a one-pair step and holds it to the neutrality class stated here:

    'bitwise'  the outputs are bit-identical with and without it (a scheduling / tiling / launch-shape choice)
    'parity'   another kernel or summation order computes the same operator: outputs agree within the parity bars of DESIGN.md
               section 5 (match sets identical off the decision margin, regression outputs 1e-3), not bit for bit
    'n/a'      does not touch results (selects a library file, a comparison leg of bench.py, a build flavour)

All of them are development / A-B aids: the defaults are the product configuration, and no deployment needs to set any.
The tuning keys are PROCESS-GLOBAL state of libfar_hip.so (std::atomic<int> g_tuning[16], far_amd/csrc/abi.hip): the one piece
of global state behind the C ABI besides the per-device one-time kernel attribute setup; include/far_hip.h says so.
"""
import os
from collections import namedtuple

Switch = namedtuple('Switch', 'env module attr off_value neutral scope doc')
# env: set to 1 to switch the feature OFF (one opt-in exception, off_value True: FAR_FPN_STREAM=1 switches it ON); module.attr: the Python flag it initialises (tests flip the attribute directly);
# scope: 'inference' (the headline step), 'training' (BASELINE configs[2]), 'bench' (bench.py only), 'load' (library selection)
SWITCHES = [
    Switch('FAR_NO_WINO', 'far_amd.ops', 'USE_WINO', False, 'parity', 'inference',
           'stride-1 3x3 convolutions on K9 (direct implicit GEMM) instead of K17 (Winograd F(2x2,3x3)); same operator, other summation'),
    Switch('FAR_NO_PAD', 'far_amd.loftr.backbone', 'PAD_CHANNELS', False, 'bitwise', 'inference',
           'the 196-channel backbone maps stored with 196 channels instead of 208 (zero weights for the extra ones: exact zeros)'),
    Switch('FAR_NO_PREFETCH', 'far_amd.loftr.model:LoFTR', 'head_prefetch', False, 'bitwise', 'inference',
           "the head's feature stage computed inside forward_rt_prediction instead of enqueued behind K1"),
    Switch('FAR_NO_SIDE_STREAM', 'far_amd.loftr.model:LoFTR', 'head_side_stream', False, 'bitwise', 'inference',
           "the head's feature stage on the step's own stream behind K1 (rounds 2-5) instead of on a second stream next to K1 and the fine level"),
    Switch('FAR_FPN_STREAM', 'far_amd.loftr.model:LoFTR', 'fpn_side_stream', True, 'bitwise', 'inference',
           "OPT-IN (1 switches it ON; default off): the FPN's fine branch on a side stream next to the coarse transformer and K1 (-0.6 ms per 32 "
           "pairs; off by default because the roofline launch -- K17 208 -> 208 @ 240x320 -- then shares the GPU and its in-step duration "
           "in a kernel trace is no longer its own)"),
    Switch('FAR_NO_GATHER_FUSE', 'far_amd.loftr.stages:FinePreprocess', 'fused_gather', False, 'bitwise', 'inference',
           'FinePreprocess.merge_feat on a materialised window tensor (K3 gather + K9) instead of K9 reading through the match indices'),
    Switch('FAR_NO_KV', 'far_amd.loftr.transformer:LoFTREncoderLayer', 'fused_kv', False, 'parity', 'inference',
           "k, v projections + K5's first half as separate launches instead of K9's k|v epilogue (the K'^T V sum changes order)"),
    Switch('FAR_NO_QAPPLY', 'far_amd.loftr.transformer:LoFTREncoderLayer', 'fused_apply', False, 'parity', 'inference',
           "q projection + K5's second half as separate launches instead of K9's attention-apply epilogue (q no longer stays in split operands)"),
    Switch('FAR_NO_STACK', 'far_amd.loftr.transformer:LocalFeatureTransformer', 'stack_self', False, 'bitwise', 'inference',
           "the two 'self' calls of an encoder layer on the two images separately instead of one call on both"),
    Switch('FAR_NO_BN', 'far_amd.ops', 'USE_HIP_BATCHNORM_TRAIN', False, 'parity', 'training',
           'training-mode BatchNorm on nn.BatchNorm2d + torch activations instead of K19 (tests/test_train_kernels_gpu.py holds K19 to the module)'),
]

ENV_ONLY = {
    'FAR_HIP_LIB': ('load', 'path of another build of libfar_hip.so (tools/ab_build.py: same-box A/B against a git revision); skips the build-id check'),
    'FAR_TUNING': ('load', 'comma-separated key=value pairs handed to far_set_tuning at load time, e.g. FAR_TUNING="10=1,8=1" (keys below)'),
    'FAR_SKIP_ASM_CHECK': ('build', "1: far_amd/build.py does not scan the generated code (K9's asm pixel loads; the LDS-DMA ring rule of every kernel with global_load_lds; the half-register writes of the asm fp16 split) after recompiling"),
    'FAR_EXTRA_HIPCC_FLAGS': ('build', 'extra hipcc flags of an experiment build (-DFAR_WINO_EXP=..., tools/wino_exp.sh); part of the build id'),
    'FAR_COMMIT': ('tools', 'commit stamp tools/collect_profiles.sh / tools/step_floors.py write into the profile files'),
    'FAR_C3_PY_NODE': ('bench', "bench.py --workload c3: the encoder layer's autograd node driven from Python instead of far_enc_layer_fwd / _bwd"),
    'FAR_C3_NO_OVERLAP': ('bench', 'bench.py --workload c3: the layer node without side streams'),
    'FAR_C3_PER_OP': ('bench', 'bench.py --workload c3: one autograd node per operator instead of per layer'),
    'FAR_TORCH_ADAMW': ('bench', 'bench.py --workload c3: torch.optim.AdamW instead of K20'),
    'FAR_CUDNN_BENCHMARK': ('bench', 'bench.py: torch.backends.cudnn.benchmark for the vendor comparison legs'),
}

Tuning = namedtuple('Tuning', 'key default values neutral doc')
TUNING = [
    Tuning(0, 3, (0, 1, 2, 4, 7), 'bitwise', 'bit mask of exact-f32 kernels that stagger wave-slot priorities: 1 k_stats, 2 k_match, 4 k_emm_pv'),
    Tuning(1, 0, (1,), 'parity', 'K1 exact-f32 variant: the other tile shape (another summation order of the row statistics)'),
    Tuning(2, 0, (1,), 'bitwise', 'conf_matrix writer variant A (store pattern)'),
    Tuning(3, 0, (1,), 'bitwise', 'conf_matrix writer variant B (store pattern)'),
    Tuning(4, 0, (1,), 'bitwise', 'K9 without the seven-tile mode / K5 windows on the generic path'),
    Tuning(5, 0, (1, 2, 4), 'bitwise', 'K5 apply: tiles per unit'),
    Tuning(6, 0, (64, 256), 'parity', "K5: tokens per K'^T V chunk (another partial-sum grouping)"),
    Tuning(7, 0, (1,), 'bitwise', 'K9 Linear launches always on full-height tiles'),
    Tuning(8, 0, (1,), 'bitwise', 'K17 splits its operands with the five-instruction split2 instead of v_fma_mix (same values)'),
    Tuning(9, 0, (1,), 'bitwise', 'K17 runs a short last channel block on the full body'),
    Tuning(10, 0, (1,), 'bitwise', "K1's match pass without the tile prescreen"),
    Tuning(11, 0, (1, 2), 'bitwise', "K14's pipeline: 0 = 4 waves / 3 slots / counted waits, two workgroups per CU; 1 = 8 waves / 4 slots / counted, one per CU (round 5's form); 2 = 4 waves / 3 slots / vmcnt(0) per phase (fallback without a wait table)"),
    Tuning(12, 0, (1,), 'parity', "K10's inference form on the exact-f32 matrix instruction instead of split fp16"),
    Tuning(15, 0, (1,), 'bitwise', 'K17: 1 = the multiplying wave group does NOT raise its issue priority (the round-4 kernel)'),
    Tuning(13, 0, (1,), 'bitwise', "K9's FPN-merge epilogue in its generic form everywhere"),
]


def off(env):
    """True when the environment switches the feature off (FAR_NO_X=1).  Only names of the registry are accepted."""
    if env not in {s.env for s in SWITCHES}:
        raise KeyError(f'{env} is not a registered far_amd switch (far_amd/flags.py)')
    return os.environ.get(env, '0') not in ('', '0')


def value(env, default=None):
    """The raw value of a registered non-boolean variable (FAR_HIP_LIB, FAR_TUNING, ...)."""
    if env not in ENV_ONLY:
        raise KeyError(f'{env} is not a registered far_amd environment variable (far_amd/flags.py)')
    return os.environ.get(env, default)


def known():
    return {s.env for s in SWITCHES} | set(ENV_ONLY)


def unknown_in_environment():
    """FAR_* names present in the environment that nothing reads (a typo of a switch would otherwise be silently ignored)."""
    skip = ('FAR_WINO_', 'FAR_K9_')            # compile-time macros of the experiment builds sometimes exported by their scripts
    return sorted(k for k in os.environ if k.startswith('FAR_') and k not in known() and not k.startswith(skip))


def target(sw):
    """(object, attribute) a Switch initialises."""
    import importlib
    mod, _, cls = sw.module.partition(':')
    obj = importlib.import_module(mod)
    return (getattr(obj, cls) if cls else obj), sw.attr
