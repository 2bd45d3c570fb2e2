#!/usr/bin/env python
"""bench.py -- image-pairs/sec of FAR's pose-estimation hot path (match + solve + regress) at 640x480.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver launches it under
torch.distributed.run with one rank per GPU (WORLD_SIZE / RANK / LOCAL_RANK in the environment); when it is started
plainly with --gpus N > 1 it launches those N ranks itself (far_amd.parallel.launch_ranks: child processes under
torch.distributed.run, started before this process touches the GPU) and returns their exit code.
One "step" = one pass of the full evaluation path
(far_amd.pipeline.test_step == lightning_loftr.py:325-343) over one batch of 32 synthetic pairs per GPU
(BASELINE.json configs[1]: "Matterport3D eval, batch 32 pairs @ 640x480, 1xMI355X"), inputs resident in HBM.
Pairs are independent: ranks shard them with no data-path collective ("weak" scaling); the only collective is
the MAX over ranks of the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAIRS_PER_GPU = 32
L = S = 4800
C = 256
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK_TFLOPS = 157.3     # dense f32-input MFMA peak (same guide)
F16_MFMA_PEAK_TFLOPS = 2500.0    # dense f16/bf16 MFMA peak (same guide; AMD headline figures include 2:1 sparsity)


from far_amd import flags as _flags  # noqa: E402  (the registry of FAR_* switches)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--pairs', type=int, default=None, help='pairs per GPU per step (default: 32 for c2, 256 for c4)')
    ap.add_argument('--vendor-train', action='store_true',
                    help='c3 only: run the training step on the differentiable vendor-op forms (tests/vendor_ops.py) instead '
                         'of the HIP forward/backward kernels -- the comparison leg')
    ap.add_argument('--workload', default='c2', choices=['c2', 'c3', 'c4', 'c5'],
                    help="c2 = BASELINE configs[1] (the headline metric: match + solve + regress, batch 32); "
                         "c4 = BASELINE configs[3] (cached-LoFTR path: GPU solver on cached correspondences + head, batch 256)")
    ap.add_argument('--hyp', type=int, default=2048, help='RANSAC hypotheses per pair (metrics.py:120)')
    ap.add_argument('--minimal', type=int, default=8, choices=[8, 5],
                    help="c2: minimal solver of the RANSAC hypotheses: 8 = the normalized 8-point north_star names (default), 5 = Nister's "
                         "five-point for every pair -- the solver class the reference actually executes (OpenCV's five-point, "
                         "ransac.py:151-157)")
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'fp16-fine', 'fp16', 'mixed16'],
                    help='backbone convolution arithmetic (far_amd.loftr.LoFTR.set_precision): fp32 = split-fp16 operand pairs (fp32-grade)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--skip-rooflines', action='store_true',
                    help='c2 child legs of other_workloads only: no isolated kernel timings / roofline object (the parent line carries them)')
    ap.add_argument('--no-other-modes', action='store_true', help='skip the informational fp16-operand leg of the default run')
    ap.add_argument('--no-other-workloads', action='store_true',
                    help='skip the short legs of the other BASELINE configs (c3, c4, c5) and the one-pair latency leg of the default run')
    ap.add_argument('--cpu-pairs', type=int, default=3)
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend of the N > 1 run ('nccl' is RCCL on ROCm; 'gloo' only for the CPU self-test)")
    ap.add_argument('--dist-at-world-1', action='store_true',
                    help="join the process group even when WORLD_SIZE is 1 (a single rank, under torch.distributed.run "
                         "--nproc-per-node 1 or plain python): init_process_group('nccl', device_id=...) = RCCL comes up, the barrier / "
                         "max-over-ranks / per-rank gather run as collectives on device tensors, and the training workload builds "
                         "SyncBatchNorm + DistributedDataParallel over it and all-reduces its gradients (to itself).  The one-GPU box's "
                         "way of executing the code path the 8-GPU run takes.")
    ap.add_argument('--share-gpu', action='store_true',
                    help="functional check of the N > 1 path on a one-GPU box: every rank uses cuda:0 (needs --backend gloo; RCCL "
                         "refuses two ranks on one device).  The timings of such a run mean nothing.")
    ap.add_argument('--selftest-launcher', action='store_true',
                    help='rendezvous + barrier + timing reduction only (no GPU work): exercises the N > 1 launch path on CPU')
    return ap.parse_args()


def _build_id():
    """far_build_id() of the loaded library = sha256/16 of far_amd/csrc + flags it was built from (checked against the sources at load)."""
    from far_amd import _lib
    return _lib.load().far_build_id().decode()


def event_time_ms(fn, iters=5, warm=2):
    """Average duration of fn() measured with HIP events on the stream the kernels are launched on
    (torch's current stream, which is the stream handed to every far_* call)."""
    for _ in range(warm):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_rooflines(n_pairs):
    """Isolated timings of the hand-written kernels at the bench shapes -> roofline entries.
    Algorithmic work per image pair:
      K1 stats / K1 match : 2*L*S*C flops each = 11.80 GFLOP (one correlation pass)
      K2 stats            : 2 dirs * 4 heads * 2*N*N*64   = 23.59 GFLOP
      K2 pv               : 23.59 (scores) + 2 dirs*4 heads*2*N*N*70 (P v~) = 49.4 GFLOP
    """
    from far_amd import ops
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(1)
    f0 = 1.2 * torch.randn(n_pairs, L, C, device=dev, generator=g)
    f1 = f0[:, torch.randperm(L, device=dev, generator=g)] + 0.1 * torch.randn(n_pairs, L, C, device=dev, generator=g)
    out = {}
    t = event_time_ms(lambda: ops.dual_softmax_stats(f0, f1, 16.0, 0.1, 1.0))
    fl = 2.0 * L * S * C * n_pairs
    out['k_stats_f32[K1, exact-f32 variant]'] = dict(ms=t, tflops=fl / t / 1e9, frac=fl / t / 1e9 / F32_MFMA_PEAK_TFLOPS)
    t2 = event_time_ms(lambda: ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0))
    out['far_coarse_match_f32[K1 all passes, exact-f32 variant]'] = dict(ms=t2, tflops=fl / t2 / 1e9, frac=fl / t2 / 1e9 / F32_MFMA_PEAK_TFLOPS)
    tm = event_time_ms(lambda: ops.coarse_match(f0[:8], f1[:8], 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, want_conf=True))
    by = (4.0 * L * S + 4.0 * (L + S) * C) * min(8, n_pairs)
    out['far_coarse_match_f32[K1 materialising conf_matrix, 8 pairs, exact-f32 variant]'] = dict(ms=tm, gbs=by / tm / 1e6, frac=by / tm / 1e6 / HBM_PEAK_GBS)
    tb = event_time_ms(lambda: ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, bf16=True))
    out['far_coarse_match_bf16[K1 all passes, bf16 MFMA variant]'] = dict(ms=tb, tflops=fl / tb / 1e9, frac=fl / tb / 1e9 / 2500.0)
    tbm = event_time_ms(lambda: ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, want_conf=True, bf16=True), iters=3, warm=1)
    byb = (4.0 * L * S + 4.0 * (L + S) * C) * n_pairs
    out['far_coarse_match_bf16[K1 materialising conf_matrix, bf16 MFMA variant]'] = dict(ms=tbm, gbs=byb / tbm / 1e6, frac=byb / tbm / 1e6 / HBM_PEAK_GBS)
    ts = event_time_ms(lambda: ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, variant='f16s'))
    out['far_coarse_match_f16s[K1 all passes, split-fp16 (default)]'] = dict(ms=ts, tflops=fl / ts / 1e9, frac=fl / ts / 1e9 / F16_MFMA_PEAK_TFLOPS)
    tsm = event_time_ms(lambda: ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, want_conf=True, variant='f16s'), iters=3, warm=1)
    out['far_coarse_match_f16s[K1 materialising conf_matrix, split-fp16]'] = dict(ms=tsm, gbs=byb / tsm / 1e6, frac=byb / tsm / 1e6 / HBM_PEAK_GBS)
    # the HBM-bound conf_matrix writer (dual_softmax_conf_f16.hip): stage 2 of far_conf_matrix_f16s timed on its own
    # (k1_conf + k1_conf_fix; operand planes and statistics prepared once by stage 1), then the whole call
    from far_amd import _lib as _l
    import ctypes as _ct
    lib_ = _l.load()
    ws_ = torch.empty(lib_.far_coarse_match_f16s_workspace_bytes(n_pairs, L, S, C), dtype=torch.uint8, device=dev)
    conf_ = torch.empty(n_pairs, L, S, device=dev)
    info_ = torch.zeros(2, dtype=torch.int32, device=dev)
    st_ = _ct.c_void_p(torch.cuda.current_stream().cuda_stream)

    def writer(stages):
        rc = lib_.far_conf_matrix_f16s(f0.data_ptr(), f1.data_ptr(), n_pairs, L, S, C, _ct.c_float(0.1), None, None, stages,
                                       conf_.data_ptr(), info_.data_ptr(), ws_.data_ptr(), None, st_)
        assert rc == 0, rc
    writer(1)
    tw = event_time_ms(lambda: writer(2), iters=5, warm=2)
    listed = int(info_[0].item())
    byw = 4.0 * L * S * n_pairs + 2.0 * (L + S) * C * n_pairs          # matrix written + fp16 operand planes read once
    out['k1_conf+k1_conf_fix[K1 conf_matrix writer alone, plain-fp16 scores + exact fix-up]'] = dict(
        ms=tw, gbs=byw / tw / 1e6, frac=byw / tw / 1e6 / HBM_PEAK_GBS, exact_entries_per_row=listed / (n_pairs * L))
    ta = event_time_ms(lambda: writer(3), iters=3, warm=1)
    out['far_conf_matrix_f16s[K1 materialising conf_matrix, all passes: planes + statistics + writer]'] = dict(
        ms=ta, gbs=byb / ta / 1e6, frac=byb / ta / 1e6 / HBM_PEAK_GBS)
    del ws_, conf_
    del f0, f1
    Z = n_pairs * 8
    q = torch.randn(Z, L, 64, device=dev, generator=g)
    k = torch.randn(Z, L, 64, device=dev, generator=g)
    v = torch.randn(Z, L, 64, device=dev, generator=g)
    pos = torch.rand(L, 6, device=dev, generator=g)
    T = torch.empty(Z, L, 70, device=dev)
    from far_amd import _lib
    lib = _lib.load()
    import ctypes
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl4 = (2.0 * L * L * 64 + 2.0 * L * L * 70) * Z
    ws = torch.empty(lib.far_emm_pv_f16s_workspace_bytes(Z, L), dtype=torch.uint8, device=dev)

    def pv16():
        lib.far_emm_pv_f16s(q.data_ptr(), k.data_ptr(), v.data_ptr(), pos.data_ptr(), Z, L, 64, ctypes.c_float(0.125),
                            1, 0, L * 64, 0, ws.data_ptr(), T.data_ptr(), None, st)
    t5 = event_time_ms(pv16, iters=3, warm=1)
    out['far_emm_pv_f16s[K2 all passes, split-fp16]'] = dict(ms=t5, tflops=fl4 / t5 / 1e9, frac=fl4 / t5 / 1e9 / F16_MFMA_PEAK_TFLOPS)
    del ws
    t3 = event_time_ms(lambda: ops.dual_softmax_stats(q, k, 1.0, 1.0, 0.125), iters=3, warm=1)
    fl3 = 2.0 * L * L * 64 * Z
    out['k_stats_f32[K2, exact-f32 variant]'] = dict(ms=t3, tflops=fl3 / t3 / 1e9, frac=fl3 / t3 / 1e9 / F32_MFMA_PEAK_TFLOPS)
    rs, cs = ops.dual_softmax_stats(q, k, 1.0, 1.0, 0.125)

    def pv():
        lib.far_emm_pv_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), pos.data_ptr(), Z, L, 64, ctypes.c_float(0.125),
                           rs.data_ptr(), cs.data_ptr(), T.data_ptr(), st)
    t4 = event_time_ms(pv, iters=3, warm=1)
    out['k_emm_pv_f32[K2, exact-f32 variant]'] = dict(ms=t4, tflops=fl4 / t4 / 1e9, frac=fl4 / t4 / 1e9 / F32_MFMA_PEAK_TFLOPS)
    del q, k, v, T
    # K9: split-fp16 implicit GEMM.  tflops = ALGORITHMIC convolution flops (2 N H W Cin Cout 9) / time, priced
    # against the dense f16 MFMA peak; the kernel issues three MFMAs per fp32-grade product, so its matrix-pipe
    # occupancy is mfma_issue_frac = 3 x frac.
    # K17 (the inference step's stride-1 3x3 layers since round 4): Winograd F(2x2, 3x3) on the same split operands -- 16 products
    # per 2x2 outputs where the direct form has 36, so it EXECUTES 3 * 16/36 = 1.33 MFMA flops per algorithmic flop (times
    # the padding of Cin to 16, Cout to 64 and the map to 16 x 16-pixel workgroup tiles); `tflops` stays algorithmic.
    nimg = 2 * n_pairs
    use_wino = ops.USE_WINO
    for label, (H, W, ci, co, ks) in {'k_conv[K9 3x3 196->196 @240x320]': (240, 320, 196, 196, 3),
                                      'k_conv[K9 3x3 128->128 @240x320]': (240, 320, 128, 128, 3),
                                      'k_conv[K9 3x3 256->256 @120x160]': (120, 160, 256, 256, 3),
                                      'k_conv[K9 linear 256->256, 4800 rows/image]': (1, 4800, 256, 256, 1)}.items():
        x = torch.randn(nimg, H, W, ci, device=dev, generator=g).relu_()
        wt = torch.randn(co, ci, ks, ks, device=dev, generator=g) * (2.0 / (ci * ks * ks)) ** 0.5
        pc = ops.PackedConv(wt, torch.ones(co, device=dev), torch.zeros(co, device=dev))
        ops.USE_WINO = False
        t9 = event_time_ms(lambda: ops.conv_nhwc(x, pc, act='relu'), iters=3, warm=1)
        ops.USE_WINO = use_wino
        fl9 = 2.0 * nimg * H * W * ci * co * ks * ks
        # MFMAs the launch executes: output channels padded to the 128-wide channel block, input channels to 16 per tap
        pad = (-(-co // 128) * 128 / co) * (-(-ci // 16) * 16 / ci)
        out[label] = dict(ms=t9, tflops=fl9 / t9 / 1e9, frac=fl9 / t9 / 1e9 / F16_MFMA_PEAK_TFLOPS,
                          mfma_issue_frac=3 * fl9 / t9 / 1e9 / F16_MFMA_PEAK_TFLOPS, mfma_executed_tflops=3 * pad * fl9 / t9 / 1e9)
        if ks == 3:
            # in the layout the step uses: channel counts padded to multiples of 16 in HBM (196 -> 208, zero weights for the rest)
            cip, cop = -(-ci // 16) * 16, -(-co // 16) * 16
            wp = torch.zeros(cop, cip, 3, 3, device=dev)
            wp[:co, :ci] = wt
            pw = ops.PackedWino(wp, torch.ones(cop, device=dev), torch.zeros(cop, device=dev))
            if cip != ci:
                xp = torch.zeros(nimg, H, W, cip, device=dev)
                xp[..., :ci] = x
                x = xp
            t17 = event_time_ms(lambda: ops.conv3x3_wino(x, pw, act='relu'), iters=3, warm=1)
            padw = (-(-co // 64) * 64 / co) * (-(-ci // 16) * 16 / ci) * (-(-H // 16) * 16 / H) * (-(-W // 16) * 16 / W)
            ex = 3.0 * 16.0 / 36.0 * padw
            out[label.replace('k_conv[K9', 'k_wino[K17')] = dict(
                ms=t17, tflops=fl9 / t17 / 1e9, frac=fl9 / t17 / 1e9 / F16_MFMA_PEAK_TFLOPS,
                mfma_issue_frac=ex * fl9 / t17 / 1e9 / F16_MFMA_PEAK_TFLOPS, mfma_executed_tflops=ex * fl9 / t17 / 1e9,
                speedup_over_k9=t9 / t17)
        del x
    # The attention half of a d_model-256 encoder layer (round 4): K9's Linear mode ending in LinearAttention -- the k | v projection
    # leaves the K'^T V state, the q projection the attention message; q, k, v never reach HBM.  Priced against HBM (the bytes the
    # three launches have to move: the two inputs, the message, the partial states) next to the unfused five launches.
    s_tok = 4800
    xs = torch.randn(nimg, s_tok, 256, device=dev, generator=g)
    wq, wk, wv = (torch.randn(256, 256, device=dev, generator=g) / 16 for _ in range(3))
    pq, pkv = ops.PackedConv(wq), ops.PackedConv(torch.cat([wk, wv], 0))
    pst = ops.PackedConv(ops.kv_interleaved_weight(wk, wv, 8))

    def unfused():
        qq = ops.linear_f16s(xs, pq)
        kk, vv = ops.linear_f16s(xs, pkv, out_planes=2)
        return ops.linear_attention(qq, kk, vv, 8)

    def fused():
        _, im = ops.linear_kv_state(xs, pst, s_tok, want_image=True)
        return ops.linear_q_apply(xs, pq, im, s_tok)
    tu = event_time_ms(unfused, iters=3, warm=1)
    tf = event_time_ms(fused, iters=3, warm=1)
    nb = 3.0 * xs.numel() * 4 + 2.0 * (nimg * s_tok // 64) * 256 * 33 * 4
    out['k_conv[K9 linear + LinearAttention epilogues: q / k | v projections + attention of a d_model-256 layer]'] = dict(
        ms=tf, gbs=nb / tf / 1e6, frac=nb / tf / 1e6 / HBM_PEAK_GBS, unfused_five_launches_ms=tu, speedup=tu / tf,
        tflops=3 * 2.0 * nimg * s_tok * 256 * 256 / tf / 1e9)
    del xs
    return out


def pmc_traffic(kernel_label, n_pairs, precision):
    """HBM/fabric bytes per launch of the dominant kernel, from the newest committed PMC pass
    (profiles/r*_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 read
    correction).  Counters cannot be collected inside this process, so this is a COMMITTED measurement, reported
    only when it was taken at this launch geometry (batch 32) and operand precision -- otherwise null.  The entry
    names the file it comes from, so that nobody divides it by a launch time of a different build unknowingly."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files or n_pairs != 32:
        return None
    path = files[-1]
    doc = json.load(open(path))
    if doc.get('precision', 'fp32') != precision:
        return None
    per = doc['per_launch']
    name = kernel_label.split('[')[0]
    ent = None
    for key, val in per.items():
        if val.get('label') == kernel_label:
            ent = val
            break
    if ent is None and name == 'k_conv' and 'profiles/r01' in path:          # round-1 file: keyed by template instantiation and grid
        ent = per.get(f'k_conv<3, 2, 2, 4, true, 1>|grid={2 * n_pairs * 30 * 20 * 256}')
    return None if ent is None else {'bytes': ent['total_bytes'], 'read_bytes': ent['read_bytes'],
                                     'write_bytes': ent['write_bytes'], 'precision': doc.get('precision', 'fp32'),
                                     'source': os.path.relpath(path, ROOT), 'commit': doc.get('commit')}


def rocprof_launch_ms(kernel_label):
    """Average duration of the roofline launch in the newest committed rocprofv3 kernel trace (profiles/r*_bench_fp32_kernel_trace.txt,
    the `roofline_launch_rocprof:` line tools/profile_report.py writes, or the per-grid table of the round 1-2 files), so that the
    HIP-event time measured here and the profiler's average for the same launch stand next to each other.  A COMMITTED measurement
    (another run, maybe another box and build): the entry names its file."""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_fp32_kernel_trace.txt')), reverse=True):
        txt = open(path).read()
        m = re.search(r'roofline_launch_rocprof: kernel=' + re.escape(kernel_label) + r' calls=(\d+) avg_us=([0-9.]+)', txt)
        if m is None and kernel_label == 'k_conv[K9 3x3 196->196 @240x320]' and re.search(r'r0[12]_bench', path):
            m = re.search(r'grid \(9830400, 1, 1\) calls (\d+)\s+avg ([0-9.]+) us', txt)
        if m:
            return {'ms': round(float(m.group(2)) / 1e3, 3), 'calls': int(m.group(1)), 'source': os.path.relpath(path, ROOT)}
    return None


class _BlockedTime:
    """Context: accumulates (in .blocked, seconds) the wall time the thread spends inside the calls that wait for the device -- the tensor
    methods .item / .cpu / .tolist / .numpy and the explicit synchronize calls."""

    def __enter__(self):
        import torch
        self.blocked, depth = 0.0, [0]

        def wrap(fn):
            def w(*a, **k):
                if depth[0]:
                    return fn(*a, **k)
                depth[0] += 1
                t = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    self.blocked += time.perf_counter() - t
                    depth[0] -= 1
            return w
        self.saved = [(torch.Tensor, nm, getattr(torch.Tensor, nm)) for nm in ('item', 'cpu', 'tolist', 'numpy')]
        self.saved += [(torch.cuda, 'synchronize', torch.cuda.synchronize), (torch.cuda.Event, 'synchronize', torch.cuda.Event.synchronize),
                       (torch.cuda.Stream, 'synchronize', torch.cuda.Stream.synchronize)]
        for obj, nm, fn in self.saved:
            setattr(obj, nm, wrap(fn))
        return self

    def __exit__(self, *exc):
        for obj, nm, fn in self.saved:
            setattr(obj, nm, fn)


def host_work_ms(step, n=3):
    """Host time a step costs its enqueuing thread, WITHOUT the time it spends blocked on the GPU: wall time of n steps (no final
    synchronisation) minus the wall time inside the calls that wait for the device (_BlockedTime).  What is left is Python + ctypes + the
    caching allocator -- the quantity N ranks on one host contend for.  (time.thread_time() is useless here: the HIP runtime spin-waits,
    so CPU time = wall time.)  Untimed extra steps."""
    import torch
    torch.cuda.synchronize()
    with _BlockedTime() as b:
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        wall = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1000.0 * (wall - b.blocked) / n


SETTLE_EXTRA = 360       # priming steps settle() may add beyond the requested --warmup (stated constant); the bound that
                         # binds is max_seconds (30 s for the headline workload, 10 s for the others): a settled box leaves after three steps, a box whose host is still paging the image in
                         # (45 ms of host work per step against 5: 98-117 ms steps) primes until the host has caught up or 30 s have passed


def settle(step, warmup=0, device='cpu', max_seconds=10.0, host_frac=None):
    """Untimed priming steps until the step time has settled (the last three within 2 % of each other and of the best seen; with
    `host_frac` -- the headline workload, whose steady state is GPU-bound -- also: host work below that fraction of the step): the
    first process on a fresh box runs its first 10-60 s with a slow host (page-in of the image; 113-117 ms steps were measured there
    against 83-85 ms in the next process on the same box -- round 6's profile run caught one such line: 274 pairs/s, then 383 under the
    profiler a minute later).  Bounded: at most `warmup + SETTLE_EXTRA` steps / `max_seconds`; a settled box leaves after three steps.  The stop decision is
    COLLECTIVE when a process group is up: `step` may contain collectives (DDP gradient all-reduce, SyncBatchNorm), so
    every rank all-reduces its continue flag (MAX) after every step and all ranks leave the loop after the same step count.
    Returns the number of steps run -- reported as `prime_steps` next to `prime_cap`, never timed."""
    import torch
    from far_amd import parallel
    cap = warmup + SETTLE_EXTRA
    ts, t_begin = [], time.perf_counter()
    while True:
        torch.cuda.synchronize()
        with _BlockedTime() as b:
            t0 = time.perf_counter()
            step()
            host = time.perf_counter() - t0 - b.blocked          # this step's host work (host_work_ms on one step)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        # ... and the host must not be the bottleneck: a box still paging the image in runs CONSISTENTLY slow steps (45 ms of host work
        # against 5), which the time criterion alone would accept as settled
        settled = len(ts) >= 3 and max(ts[-3:]) < 1.02 * min(ts[-3:]) and min(ts[-3:]) < 1.02 * min(ts) and (host_frac is None or host < host_frac * ts[-1])
        go_on = 0.0 if (settled or len(ts) >= cap or time.perf_counter() - t_begin >= max_seconds) else 1.0
        if parallel.max_over_ranks(go_on, device=device) <= 0.0:          # (a plain float at world size 1 without a group)
            break
    return len(ts)


def other_workloads(a):
    """Short legs of the other BASELINE configs and of the reference's deployment shape (one pair per step), each as a child process
    started AFTER this process has finished its own timing (python bench.py --workload ...: its own line, parsed here), so that
    every workload number is in the line the driver records.  Informational: never `value`."""
    import subprocess
    legs = [('c3_training_step', ['--workload', 'c3', '--steps', '12', '--warmup', '3']),
            ('c4_cached_path', ['--workload', 'c4', '--steps', '10', '--warmup', '2']),
            ('c4_cached_path_fp16', ['--workload', 'c4', '--precision', 'fp16', '--steps', '10', '--warmup', '2']),
            ('c5_mapfree_544x720', ['--workload', 'c5', '--steps', '10', '--warmup', '3']),
            ('c5_mapfree_544x720_fp16', ['--workload', 'c5', '--precision', 'fp16', '--steps', '10', '--warmup', '3']),
            ('c2_one_pair_latency', ['--pairs', '1', '--steps', '20', '--warmup', '5', '--skip-rooflines']),
            ('c2_five_point_solver', ['--minimal', '5', '--steps', '10', '--warmup', '3', '--skip-rooflines'])]
    out = {}
    for name, args in legs:
        cmd = [sys.executable, os.path.abspath(__file__)] + args + ['--hyp', str(a.hyp), '--no-cpu-baseline', '--no-other-modes',
                                                                  '--no-other-workloads']
        t0 = time.perf_counter()
        try:
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
            lines = [l for l in pr.stdout.splitlines() if l.startswith('{')]
            if pr.returncode != 0 or not lines:
                out[name] = {'error': f'rc {pr.returncode}', 'stderr_tail': pr.stderr[-300:]}
                continue
            r = json.loads(lines[-1])
            ent = {'value': r['value'], 'unit': r['unit'], 'ms_per_step': r['ms_per_step'], 'steps': r['steps'], 'warmup': r['warmup'],
                   'pairs_per_step': r['config'].get('pairs_per_gpu'), 'metric': r['metric'], 'wall_s': round(time.perf_counter() - t0, 1)}
            for k in ('k4', 'k1', 'vendor_convolution'):
                if k in r:
                    ent[k] = r[k]
            if name == 'c4_cached_path_fp16':
                ent['note'] = "the same leg under LoFTR.set_precision('fp16') (16-bit operands in the head's K9 / K2 launches); informational"
            if name == 'c5_mapfree_544x720_fp16':
                ent['note'] = ("BASELINE configs[4] names 'fp16 MFMA': the same leg under LoFTR.set_precision('fp16') (16-bit operands in K9 / K1, "
                               "fp32 tensors and accumulation); the leg above is the fp32-grade form")
            if name == 'c2_five_point_solver':
                ent['pose_error'] = r['config'].get('pose_error')
                ent['note'] = ("the headline workload with Nister's five-point solver as the minimal solver of every pair (H = 2048 models "
                               "from H / 10 samples): the solver CLASS the reference executes (OpenCV's five-point on the host, "
                               "ransac.py:151-157, cv_geometry.py:836-859); the headline uses the normalized 8-point north_star names")
            if name == 'c2_one_pair_latency':
                ent['note'] = ('the reference\'s deployment shape (scripts/eval_matterport.sh: batch_size 1): ms_per_step IS the latency of one '
                               'pair through match + 2 solver rounds + 2 head calls')
            out[name] = ent
        except Exception as e:                      # a leg must never take the headline line down
            out[name] = {'error': repr(e)[:300]}
    return out


def cpu_baseline(n_pairs, hyp):
    """The oracle (CPU restatement of the reference path, oracle/model.py) timed on this box's host cores: the same
    synthetic workload, all cores (capped at 32) on `n_pairs` pairs and ONE thread on one pair, each split into the
    metric's three stages (match / solve x2 / regress x2).  A reported baseline, not the optimisation target."""
    import json as _json
    from far_amd import synth
    from far_amd.config import far_eval_config
    from oracle import head as ohead
    from oracle import model as om
    cfg = far_eval_config()
    man = _json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g8_state_dict_manifest.json')))
    w = om.Weights(synth.synthetic_state_dict({k: tuple(v) for k, v in man.items()}))
    pcl = np.random.RandomState(0).uniform(low=-3.0, high=3.0, size=(300, 3)).astype(np.float32)
    pos = ohead.positional_encodings()

    def leg(threads, pairs):
        """om.test_step's call order (lightning_loftr.py:325-343) with a clock around each stage."""
        torch.set_num_threads(threads)
        im0, im1 = synth.synth_image_pair(pairs, seed=4242)
        K = np.stack([synth.MP3D_K] * pairs)
        t = {'match': 0.0, 'solve': 0.0, 'regress': 0.0}
        t0 = time.time()
        data = om.matcher_forward(w, cfg, im0, im1)
        t['match'] = time.time() - t0
        for b in range(pairs):
            t0 = time.time()
            rt, nb, na, ti, ul, _ = om.solve_pair(data, b, K[b], K[b], cfg['solver'], None, 0, hyp, pcl)
            t['solve'] += time.time() - t0
            for i in range(2):
                t0 = time.time()
                lp, ilp = om.preprocess_helper(cfg, rt, na, nb, ti, ul)
                reg, _, _ = om.head_forward(w, cfg, data['featmap0'][b:b + 1], data['featmap1'][b:b + 1], lp, ilp, pos)
                prior = om.prior_from_regressed(reg)
                t['regress'] += time.time() - t0
                if i == 0:
                    t0 = time.time()
                    rt, nb, na, ti, ul, _ = om.solve_pair(data, b, K[b], K[b], cfg['solver'], prior, 0, hyp, pcl)
                    t['solve'] += time.time() - t0
        tot = sum(t.values())
        return {'value': round(pairs / tot, 4), 'cores': threads, 'pairs': pairs, 'wall_s': round(tot, 2),
                'stage_s_per_pair': {k_: round(v_ / pairs, 3) for k_, v_ in t.items()}}
    cores = min(os.cpu_count() or 1, 32)     # more threads only add synchronisation overhead to these CPU ops
    full = leg(cores, n_pairs)
    one = leg(1, 1)
    torch.set_num_threads(cores)
    return {'value': full['value'], 'unit': 'image-pairs/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{n_pairs} pair(s) of the same synthetic 640x480 workload through oracle/model.py in test_step\'s order '
                      f'(numpy + torch-CPU fp32, solver float64, H={hyp}), {full["wall_s"]} s wall; plus 1 pair on 1 thread',
            'stage_s_per_pair': full['stage_s_per_pair'], 'single_thread': one,
            'solver_note': 'solver stage = the oracle\'s batched normalized 8-point prior-RANSAC in numpy (the reference executes '
                           'OpenCV\'s 5-point on the host; cv2 is not in this image)'}


def mfma_sustained_peak():
    """far_mfma_probe_f16 timed on this box: the dense f16 MFMA rate the part sustains with K9's register / LDS footprint and
    nothing else in the loop (both forms; a burst long enough for the power limit to settle)."""
    import ctypes
    from far_amd import _lib
    lib = _lib.load()
    sink = torch.zeros(1, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for mode, name in ((0, 'operands_in_registers'), (1, 'operands_from_lds')):
        fl = ctypes.c_double()

        def run():
            rc = lib.far_mfma_probe_f16(mode, 16000, 1, sink.data_ptr(), ctypes.byref(fl), st)
            assert rc == 0, rc
        t = event_time_ms(run, iters=8, warm=4)
        out[name] = {'tflops': round(fl.value / t / 1e9, 1), 'ms_per_launch': round(t, 3)}
    return out


def selftest_launcher(a):
    """N > 1 launch path without GPU work: every rank joins the process group, passes a barrier, contributes a
    fake step time; rank 0 prints the JSON skeleton with the rank count the backend reports."""
    import torch.distributed as dist
    from far_amd import parallel
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group(a.backend)
    dist_world = dist.get_world_size() if world > 1 else 1
    if world > 1:
        dist.barrier()
    dt = parallel.max_over_ranks(0.001 * (rank + 1))
    per = parallel.gather_floats(0.001 * (rank + 1))
    if rank == 0:
        print(json.dumps({'selftest': 'launcher', 'n_gpus': dist_world, 'requested_gpus': a.gpus, 'backend': a.backend,
                          'max_s': dt, 'per_rank_s': per}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_c4(a, dev, world, rank, dist):
    """BASELINE configs[3]: InteriorNet-T cached-LoFTR path -- no matcher; cached transformer features + cached fine
    correspondences -> GPU solver (two rounds, the second with the head's pose as prior) + EMM head (two calls), batch 256
    (far_amd.pipeline.cached_step).  A second JSON line shape: same contract keys, metric named for this workload."""
    from far_amd import parallel
    from far_amd.config import far_eval_config
    from far_amd.loftr import LoFTR
    from far_amd.pipeline import cached_step
    from far_amd.supervision import compute_supervision_RT
    from far_amd.config import RunCfg
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from tests.util import two_view_scene
    B = a.pairs
    cfg = far_eval_config()
    cfg['from_saved_preds'] = 'loftr_preds'
    model = LoFTR(cfg).eval()
    man = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g8_state_dict_manifest.json')))
    from far_amd import synth
    sd = synth.synthetic_state_dict({k: tuple(v) for k, v in man.items() if k.startswith('loftr_regress.')})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev)
    model.set_precision(a.precision)
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    f0 = torch.randn(B, L, C, device=dev, generator=g)
    f1 = 0.5 * f0 + torch.randn(B, L, C, device=dev, generator=g)
    rng = np.random.default_rng(99 + rank)
    scenes = [two_view_scene(int(rng.integers(800, 2000)), seed=1000 * rank + b, outlier_frac=0.3) for b in range(B)]
    cnt = [len(s_[0]) for s_ in scenes]
    mk0 = torch.from_numpy(np.concatenate([s_[0] for s_ in scenes])).to(dev)
    mk1 = torch.from_numpy(np.concatenate([s_[1] for s_ in scenes])).to(dev)
    bids = torch.repeat_interleave(torch.arange(B), torch.tensor(cnt)).to(dev)
    K = torch.from_numpy(np.stack([s_[2] for s_ in scenes])).to(dev)
    base = {'featmap0': f0, 'featmap1': f1, 'mkpts0_f': mk0, 'mkpts1_f': mk1, 'm_bids': bids, 'b_ids': bids,
            'match_counts': torch.tensor(cnt), 'K0': K, 'K1': K.clone(), 'dataset_name': ['interiornet_streetlearn']}

    def step():
        batch = dict(base)
        cached_step(model, batch, H=a.hyp, seed=0)
        return batch

    prime = max(0, 3 - a.warmup)
    for _ in range(prime + a.warmup):
        last = step()
    # (host_frac only for the batched headline shape: a one-pair step is host-bound in its steady state)
    prime += settle(step, a.warmup, dev, max_seconds=30.0 if a.pairs >= 8 else 10.0, host_frac=0.3 if a.pairs >= 8 else None)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    per_rank_ms = [round(1000 * x / a.steps, 3) for x in parallel.gather_floats(dt, device=dev)]
    dt = parallel.max_over_ranks(dt, device=dev)
    if rank == 0:
        # K4 alone (one round with the head's prior): hypotheses per second
        run = RunCfg('prior_ransac', 2)
        t_solver = event_time_ms(lambda: compute_supervision_RT(dict(last), run, H=a.hyp, seed=0), iters=5, warm=2)
        from far_amd import metrics as fm
        Rgt = torch.from_numpy(np.stack([s_[3] for s_ in scenes])).to(dev)
        tgt = torch.from_numpy(np.stack([s_[4] for s_ in scenes])).to(dev)
        Tgt = torch.cat([Rgt, tgt[:, :, None]], 2)
        rt = last['loftr_rt'].reshape(-1, 3, 4)
        te, Re, _ = fm.relative_pose_error_batch(Tgt, rt[:, :, :3], rt[:, :, 3])
        auc = fm.error_auc_device(torch.maximum(te, Re))
        res = {
            'metric': 'image-pairs/sec (solve+regress on cached LoFTR predictions) -- BASELINE configs[3], not the headline metric',
            'value': round(world * B * a.steps / dt, 3), 'unit': 'image-pairs/sec', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'prime_steps': prime, 'prime_cap': max(0, 3 - a.warmup) + a.warmup + SETTLE_EXTRA, 'ms_per_step': round(1000 * dt / a.steps, 3),
            'per_rank_ms_per_step': per_rank_ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('f32 (head: split-f16x3 operands, fp32 accumulate) / f64 (solver)' if a.precision == 'fp32' else
                      f'f32 tensors/accumulation, {a.precision} matrix operands (head) / f64 (solver)'), 'data': 'synthetic',
            'config': {'workload': 'InteriorNet-T-shaped cached-LoFTR path (BASELINE configs[3]): batch 256 pairs per GPU, cached '
                                   'coarse features + 800-2000 cached correspondences per pair (30 % outliers), GPU 8-pt prior-RANSAC '
                                   'solve (x2 rounds) + EMM head (x2), seeded random weights',
                       'pairs_per_gpu': B, 'hypotheses': a.hyp, 'correspondences_per_pair': round(float(np.mean(cnt)), 1),
                       'solver_success_frac': float(last['solver_status'].float().mean().item()),
                       'solver_median_R_deg': round(float(Re.median()), 3), 'solver_median_t_deg': round(float(te.median()), 3),
                       'solver_pose_auc': {k_: round(v_, 4) for k_, v_ in auc.items()},
                       'parallelism': f'dp{world} (independent pairs, no data-path collective)'},
            'k4': {'ms_per_round': round(t_solver, 3), 'hypotheses_per_sec': round(B * a.hyp / t_solver * 1e3, 1),
                   'pairs_per_sec_solver_only': round(B / t_solver * 1e3, 1),
                   'note': 'far_solver_f64 + far_pose_pack_f64, one prior-RANSAC round of the whole batch (HIP events)'},
        }
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_c3(a, dev, world, rank, dist):
    """BASELINE configs[2]: the training step of mp3d_loftr/scripts/train_matterport.sh (last stage, "FAR (full)":
    batch_size 1 per GPU, AdamW lr 1e-5, losses coarse focal + fine l2-with-std + 6D pose L1), in the reference's order
    (far_amd.pipeline.train_step = PL_LoFTR._trainval_inference) + backward + optimizer step; under N > 1 ranks the model
    is wrapped in DistributedDataParallel (gradient all-reduce over RCCL = the one exchange step of this path)."""
    from far_amd import parallel, synth
    from far_amd.config import far_train_config, RunCfg
    from far_amd.loftr import LoFTR
    from far_amd.loftr.transformer import LoFTREncoderLayer, CrossAttention
    from far_amd.losses import LoFTRLoss
    from far_amd.pipeline import train_step
    B = a.pairs
    cfg = far_train_config()
    model = LoFTR(cfg['loftr'])
    synth.load_synthetic(model, seed=0)
    model = model.to(dev).train()
    if _flags.value('FAR_C3_PY_NODE') == '1':        # A/B aid: the layer node driven from Python instead of far_enc_layer_fwd / _bwd
        from far_amd.loftr.transformer import LoFTREncoderLayer as _L3
        _L3.native_node = False
    if _flags.value('FAR_C3_NO_OVERLAP') == '1':     # A/B aid: the layer node without side streams
        from far_amd.loftr.transformer import LoFTREncoderLayer as _L2
        _L2.overlap = False
    if _flags.value('FAR_C3_PER_OP') == '1':         # A/B aid: one autograd node per operator (round-3 mid state) instead of per layer
        from far_amd import ops as _ops_ln
        from far_amd.loftr.transformer import LoFTREncoderLayer as _L
        _L.layer_node = False
        _ops_ln.USE_HIP_LAYERNORM_TRAIN = False
    if a.vendor_train:
        # the comparison leg runs torch / vendor-library compositions that are NOT part of the package: install the test-side helper
        from far_amd import _vendor
        from tests import vendor_ops
        _vendor.install(vendor_ops)
        LoFTREncoderLayer.hip_training = False
        CrossAttention.hip_training = False
        from far_amd.loftr.backbone import ResNetFPN_8_2
        from far_amd.loftr.stages import FinePreprocess
        ResNetFPN_8_2.hip_training = False
        FinePreprocess.hip_training = False
        from far_amd import ops as _ops
        _ops.USE_HIP_WGRAD = False
        _ops.USE_HIP_LAYERNORM_TRAIN = False
        model.coarse_matching.materialize_conf = True          # dense conf_matrix through the vendor ops + autograd
    loss_fn = LoFTRLoss(cfg).train()
    fwd = model
    if dist is not None:
        from torch.nn.parallel import DistributedDataParallel as DDP
        if a.backend == 'nccl':            # the reference trains with sync_batchnorm (train.py:342, :350-352); RCCL only
            model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        # no device_ids: the module already lives on `dev`, and a DDP built WITH device_ids passes dict inputs through
        # _recursive_to, which rebuilds them -- LoFTR.forward writes its results into the dict it is given
        # (pipeline._trainval_inference merges a returned copy back, tests/test_multirank_gpu.py covers both forms)
        fwd = DDP(model)
    # src/optimizers/__init__.py:5-16, default.py TRAINER.*: AdamW; K20 (one launch over all 189 tensors) unless --vendor-train
    from far_amd.optim import AdamW as _FarAdamW
    opt = (torch.optim.AdamW if (a.vendor_train or _flags.value('FAR_TORCH_ADAMW') == '1') else _FarAdamW)(model.parameters(), lr=1e-5, weight_decay=0.1)
    # synthetic supervision: banded lateral disparities -> ground-truth coarse matches + warped grid (far_amd/synth.py)
    base = synth.synth_training_batch(B, seed=1234 + rank, device=dev)
    n_gt = int(base['spv_b_ids'].numel()) // B
    run = RunCfg('prior_ransac', 2)

    def step():
        batch = dict(base)
        train_step(model, batch, loss_fn, run, H=a.hyp, seed=0, forward=fwd)
        batch['loss'].backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return batch

    prime = max(0, 3 - a.warmup)
    for _ in range(prime + a.warmup):
        last = step()
    # (host_frac only for the batched headline shape: a one-pair step is host-bound in its steady state)
    prime += settle(step, a.warmup, dev, max_seconds=30.0 if a.pairs >= 8 else 10.0, host_frac=0.3 if a.pairs >= 8 else None)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    per_rank_ms = [round(1000 * x / a.steps, 3) for x in parallel.gather_floats(dt, device=dev)]
    per_rank_host_ms = [round(x, 3) for x in parallel.gather_floats(host_work_ms(step), device=dev)]
    dt = parallel.max_over_ranks(dt, device=dev)
    # The path's one exchange step: the gradient all-reduce.  Reported two ways: a stand-alone all-reduce of
    # one flat fp32 buffer of the gradients' size (what the ring costs when nothing overlaps it), and the part of it that is
    # EXPOSED in the step = step time with the hooks armed minus step time under no_sync() (same work, no exchange).
    exchange = None
    if dist is not None:
        n_param = sum(p.numel() for p in model.parameters() if p.requires_grad)
        flat = torch.zeros(n_param, dtype=torch.float32, device=dev)
        for _ in range(2):
            dist.all_reduce(flat)
        fence()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            dist.all_reduce(flat)
        fence()
        ar_ms = parallel.max_over_ranks(time.perf_counter() - t1, device=dev) / reps * 1000
        fence()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            with fwd.no_sync():
                step()
        fence()
        nosync_ms = parallel.max_over_ranks(time.perf_counter() - t1, device=dev) / a.steps * 1000
        nbytes = 4 * n_param
        ring = 2 * (world - 1) / world * nbytes                       # bytes each rank sends (= receives) in a ring all-reduce
        exchange = {'what': 'gradient all-reduce, fp32, DistributedDataParallel buckets overlapped with backward',
                    'bytes': nbytes, 'standalone_ms': round(ar_ms, 3), 'bus_GBps': round(ring / ar_ms / 1e6, 1),
                    'per_link_bound_ms': round(ring / 153e9 * 1000, 3),
                    'bound_note': 'ring over xGMI: 2(N-1)/N x bytes through one ~153 GB/s link per direction (MI355X_MICROARCH.md)',
                    'step_ms_without_exchange': round(nosync_ms, 3),
                    'exposed_ms_in_step': round(1000 * dt / a.steps - nosync_ms, 3),
                    'backend': dist.get_backend(), 'ranks': world}
    per_rank_peak = None
    if dist is not None:                # every rank's sustained dense-f16 MFMA rate, all ranks at once (see main())
        pk = max(v['tflops'] for v in mfma_sustained_peak().values())
        per_rank_peak = [round(x, 1) for x in parallel.gather_floats(pk, device=dev)]
    if rank == 0:
        sc = {k_: round(float(v_), 5) for k_, v_ in last['loss_scalars'].items() if k_.startswith('loss')}
        res = {
            'metric': 'image-pairs/sec (training step: forward + backward + AdamW) -- BASELINE configs[2], not the headline metric',
            'value': round(world * B * a.steps / dt, 3), 'unit': 'image-pairs/sec', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'prime_steps': prime, 'prime_cap': max(0, 3 - a.warmup) + a.warmup + SETTLE_EXTRA, 'ms_per_step': round(1000 * dt / a.steps, 3),
            'per_rank_ms_per_step': per_rank_ms, 'host_ms_per_step': max(per_rank_host_ms), 'per_rank_host_ms_per_step': per_rank_host_ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'dtype_note': 'fp32 tensors and accumulation; K1 / K9 (Linear and backbone convolutions) / K2 forward on split-f16x3 operands, their '
                          'backward kernels on fp16 (K1, K2) or split-f16x3 (K9 dgrad, K16 wgrad) operands; stem forward + wgrad exact fp32 '
                          'MFMA (K10); LayerNorm forward + backward K6; BatchNorm, interpolation, elementwise glue: vendor fp32',
            'vendor_convolution': 'none' if not a.vendor_train else 'all (comparison leg)',
            'config': {'workload': 'Matterport3D-shaped training step (BASELINE configs[2]): ' + str(B) + ' pair(s) @ 640x480 per GPU, '
                                   'matcher in training mode (sampled / padded coarse matches), solver x2, head x2 (last with grad), '
                                   'coarse focal + fine l2_with_std + 6D pose L1 losses, backward, AdamW; seeded random weights',
                       'pairs_per_gpu': B, 'hypotheses': a.hyp,
                       'training_kernels': 'vendor ops + autograd (comparison leg)' if a.vendor_train else
                                           'HIP forward+backward: K1 sparse-position conf, K5, K9 Linear + backbone convolutions (dgrad), K16 '
                                           'convolution / Linear / stem weight gradients (deterministic), K6, K3 window gather / scatter, K2; one '
                                           'autograd node per encoder layer driven by the library (far_enc_layer_fwd / _bwd, side streams), both '
                                           'images through a self-attention layer in one call; all weight images re-packed in two launches per step',
                       'gt_coarse_matches_per_pair': n_gt, 'sampled_matches': int(last['b_ids'].numel()),
                       'losses': sc,
                       'parallelism': (f'ddp{world} (gradient all-reduce over {"RCCL" if a.backend == "nccl" else a.backend})' if dist is not None
                                       else 'single GPU (no exchange step)')},
            'exchange': exchange, 'per_rank_sustained_peak': per_rank_peak,
            'process_group': {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                              'sync_batchnorm': any(isinstance(m_, torch.nn.SyncBatchNorm) for m_ in model.modules())} if dist is not None else None,
        }
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_c5(a, dev, world, rank, dist):
    """BASELINE configs[4]: Map-free-shaped matcher + solver at 544 x 720 (coarse grid 68 x 90, L = S = 6120), the batched form
    of the reference's per-sample loop (mapfree_6dreg/lib/models/regression/model.py:236-273 -> far_amd.mapfree.match_and_solve):
    LoFTR match + GPU 8-pt RANSAC, a second solver round with a pose prior (the `use_prior` loop).  No EMM head: the
    reference's head is tied to the 60 x 80 grid."""
    from far_amd import parallel, synth, ops
    from far_amd.config import far_eval_config
    from far_amd.loftr import LoFTR
    from far_amd.mapfree import EssentialMatrixSolver, match_and_solve
    B = a.pairs
    model = LoFTR(far_eval_config()).eval()
    synth.load_synthetic(model, seed=0)
    model = model.to(dev)
    model.set_precision(a.precision)
    im0, im1 = synth.synth_image_pair(B, seed=1234 + rank, hw=(544, 720))
    Kc = np.array([[590.0, 0, 360.], [0, 590.0, 272.], [0, 0, 1.]])
    K = torch.from_numpy(np.stack([Kc] * B)).to(dev)
    base = {'image0': torch.from_numpy(im0).to(dev), 'image1': torch.from_numpy(im1).to(dev), 'K_color0': K, 'K_color1': K.clone()}
    solver = EssentialMatrixSolver(None, use_prior_ransac=True, H=a.hyp, seed=0)
    prior = np.stack([np.concatenate([np.eye(3), np.array([[-1.0], [0.0], [0.0]])], 1)] * B).astype(np.float32)

    def step():
        batch = dict(base)
        match_and_solve(model, batch, solver, priorRT=None)
        # second loop of `use_prior` (model.py:239-241): the solver again, now with a prior pose
        out = solver.solve_batch(batch['mkpts0_f'], batch['mkpts1_f'], batch['match_counts'].tolist(), K, K, prior)
        batch['solver_status2'] = out['status']
        return batch

    prime = max(0, 3 - a.warmup)
    for _ in range(prime + a.warmup):
        last = step()
    # (host_frac only for the batched headline shape: a one-pair step is host-bound in its steady state)
    prime += settle(step, a.warmup, dev, max_seconds=30.0 if a.pairs >= 8 else 10.0, host_frac=0.3 if a.pairs >= 8 else None)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    per_rank_ms = [round(1000 * x / a.steps, 3) for x in parallel.gather_floats(dt, device=dev)]
    dt = parallel.max_over_ranks(dt, device=dev)
    if rank == 0:
        Lc = 68 * 90
        g = torch.Generator(device=dev).manual_seed(1)
        f0 = 1.2 * torch.randn(B, Lc, C, device=dev, generator=g)
        f1 = f0[:, torch.randperm(Lc, device=dev, generator=g)] + 0.1 * torch.randn(B, Lc, C, device=dev, generator=g)
        t_k1 = event_time_ms(lambda: ops.coarse_match(f0, f1, 0.1, 0.2, 2, (68, 90), (68, 90), 8.0, variant='f16s'))
        res = {
            'metric': 'image-pairs/sec (match+solve) at 544x720 -- BASELINE configs[4], not the headline metric',
            'value': round(world * B * a.steps / dt, 3), 'unit': 'image-pairs/sec', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'prime_steps': prime, 'prime_cap': max(0, 3 - a.warmup) + a.warmup + SETTLE_EXTRA, 'ms_per_step': round(1000 * dt / a.steps, 3),
            'per_rank_ms_per_step': per_rank_ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if a.precision == 'fp32' else f'f32 tensors/accumulation, {a.precision} matrix operands', 'data': 'synthetic',
            'config': {'workload': 'Map-free-shaped matcher + solver (BASELINE configs[4]): ' + str(B) + ' pairs @ 544x720 per GPU '
                                   '(coarse 68x90, L = S = 6120), LoFTR match + GPU 8-pt RANSAC + a prior-RANSAC round, seeded random weights',
                       'pairs_per_gpu': B, 'hypotheses': a.hyp, 'matches_per_pair': round(float(last['m_bids'].numel()) / B, 1),
                       'solver_success_frac': float(last['solver_status'].float().mean().item()),
                       'parallelism': f'dp{world} (independent pairs, no data-path collective)'},
            'k1': {'ms': round(t_k1, 3), 'pairs': B, 'L': Lc,
                   'tflops': round(2.0 * Lc * Lc * C * B / t_k1 / 1e9, 2),
                   'frac_of_f16_mfma_peak': round(2.0 * Lc * Lc * C * B / t_k1 / 1e9 / F16_MFMA_PEAK_TFLOPS, 4),
                   'note': 'far_coarse_match_f16s, all passes; algorithmic flops = 2 L S C per pair (one correlation; the kernel executes three passes of three f16 MFMAs per product), HIP events'},
        }
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if a.pairs is None:
        a.pairs = {'c4': 256, 'c3': 1, 'c5': 16}.get(a.workload, PAIRS_PER_GPU)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as children BEFORE this process initialises the GPU
        from far_amd import parallel
        sys.exit(parallel.launch_ranks(a.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    if a.selftest_launcher:
        return selftest_launcher(a)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = 0 if a.share_gpu else int(os.environ.get('LOCAL_RANK', '0'))
    if a.share_gpu and world > 1 and a.backend != 'gloo':
        sys.exit('--share-gpu needs --backend gloo')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1 or a.dist_at_world_1:
        import torch.distributed as dist
        kw = {}
        if 'MASTER_ADDR' not in os.environ or 'RANK' not in os.environ:      # plain `python bench.py --dist-at-world-1`: no launcher
            from far_amd import parallel as _par
            kw = dict(init_method=f'tcp://127.0.0.1:{_par.free_port()}', rank=0, world_size=1)
        dist.init_process_group(a.backend, device_id=dev if a.backend == 'nccl' else None, **kw)
        world = dist.get_world_size()            # the rank count the backend (RCCL) reports
    if a.workload == 'c4':
        return bench_c4(a, dev, world, rank, dist)
    if a.workload == 'c3':
        return bench_c3(a, dev, world, rank, dist)
    if a.workload == 'c5':
        return bench_c5(a, dev, world, rank, dist)
    from far_amd import synth
    from far_amd.config import far_eval_config
    from far_amd.loftr import LoFTR
    from far_amd.pipeline import test_step

    if _flags.value('FAR_CUDNN_BENCHMARK') == '1':
        torch.backends.cudnn.benchmark = True      # MIOpen exhaustive solver search (experiment switch)
    cfg = far_eval_config()
    model = LoFTR(cfg).eval()
    synth.load_synthetic(model, seed=0)
    model = model.to(dev)
    model.set_precision(a.precision)
    im0, im1 = synth.synth_image_pair(a.pairs, seed=1234 + rank)
    K = torch.from_numpy(np.stack([synth.MP3D_K] * a.pairs)).to(dev)
    base = {'image0': torch.from_numpy(im0).to(dev), 'image1': torch.from_numpy(im1).to(dev), 'K0': K, 'K1': K.clone(),
            'dataset_name': ['mp3d']}

    from far_amd.config import RunCfg as _RunCfg
    run_cfg = _RunCfg(cfg['solver'], cfg.get('fine_pred_steps', 2), minimal_solver=a.minimal)

    def step(seed=0):
        batch = dict(base)
        test_step(model, batch, run_cfg=run_cfg, H=a.hyp, seed=seed)
        return batch

    # the caching allocator still grows (multi-GB hipMallocs) during the first three steps: when fewer warm-up steps
    # are asked for, the difference runs as separate untimed priming steps (reported as `prime_steps`)
    prime = max(0, 3 - a.warmup)
    for _ in range(prime + a.warmup):
        last = step()
    # (host_frac only for the batched headline shape: a one-pair step is host-bound in its steady state)
    prime += settle(step, a.warmup, dev, max_seconds=30.0 if a.pairs >= 8 else 10.0, host_frac=0.3 if a.pairs >= 8 else None)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    from far_amd import parallel
    per_rank_ms = [round(1000 * x / a.steps, 3) for x in parallel.gather_floats(dt, device=dev)]
    per_rank_host_ms = [round(x, 3) for x in parallel.gather_floats(host_work_ms(step), device=dev)]
    dt = parallel.max_over_ranks(dt, device=dev)          # the slowest rank defines the step time
    matches = float(last['b_ids'].numel()) / a.pairs
    ok_frac = float(last['solver_status'].float().mean().item())
    # pose error of the solver and of the blended head output against the synthetic ground truth
    # (pure lateral translation, no rotation): informational -- the banded synthetic pairs carry the usual
    # small-baseline translation/rotation ambiguity; accuracy on Matterport needs the real checkpoint + data.
    def pose_errors(out):
        from far_amd import metrics as fm
        from far_amd.pose6d import pose_mean_6d, pose_std_6d, rotation_6d_to_matrix
        T_gt = torch.eye(4, dtype=torch.float64, device=dev).repeat(a.pairs, 1, 1)
        T_gt[:, 0, 3] = -1.0
        rt = out['loftr_rt'].reshape(-1, 3, 4)
        te, Re, _ = fm.relative_pose_error_batch(T_gt, rt[:, :, :3], rt[:, :, 3])
        reg = out['regressed_rt'].detach().float().cpu()
        Rr = rotation_6d_to_matrix(reg[:, 3:] * pose_std_6d[3:] + pose_mean_6d[3:])
        tr = reg[:, :3] * pose_std_6d[:3] + pose_mean_6d[:3]
        te2, Re2, _ = fm.relative_pose_error_batch(T_gt.cpu(), Rr, tr)
        return {'solver_median_R_deg': round(float(Re.median()), 3), 'solver_median_t_deg': round(float(te.median()), 3),
                'head_median_R_deg': round(float(Re2.median()), 3), 'head_median_t_deg': round(float(te2.median()), 3)}
    pose_err = dict(pose_errors(last), note='vs synthetic GT (R=I, t=-x); random-weight head; Matterport accuracy pending (no ckpt/data offline)')

    # N > 1: the dense-f16 MFMA rate every rank sustains right now (far_mfma_probe_f16 on all ranks at once): under one node's
    # power envelope a sub-linear scaling curve can then be told apart from ranks that simply clock lower together
    per_rank_peak = None
    if world > 1:
        pk = max(v['tflops'] for v in mfma_sustained_peak().values())
        per_rank_peak = [round(x, 1) for x in parallel.gather_floats(pk, device=dev)]
    if rank == 0:
        kr, roof = {}, None
        if not a.skip_rooflines:
            kr = kernel_rooflines(a.pairs)
            # dominant kernel of the step: the backbone's 3x3 convolutions (45 % of the kernel-busy time, profiles/); its costliest launch
            # is reported -- K17 (Winograd) since round 4, K9 (direct) when ops.USE_WINO is off
            from far_amd import ops as _ops
            wino = _ops.USE_WINO
            dom = 'k_wino[K17 3x3 196->196 @240x320]' if wino else 'k_conv[K9 3x3 196->196 @240x320]'
            tr = pmc_traffic(dom, a.pairs, a.precision)
            probe = mfma_sustained_peak()
            sustained = max(v['tflops'] for v in probe.values())
            executed = kr[dom]['mfma_executed_tflops']
            per_alg = 3.0 * 16.0 / 36.0 if wino else 3.0          # MFMA flops issued per algorithmic flop, before padding
            roof = {'kernel': dom, 'bound': 'mfma', 'achieved': round(kr[dom]['tflops'], 2), 'peak': F16_MFMA_PEAK_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(kr[dom]['frac'], 4),
                    # what the part sustains under dense f16 MFMA issue, measured on THIS box just now (far_mfma_probe_f16: K9's
                    # accumulator / fragment footprint, random operands, nothing else in the loop): the nominal 2.5 PFLOP/s assumes
                    # 2.4 GHz, the power-limited clock under this load is 1.5-1.8 GHz
                    'sustained_peak': sustained, 'sustained_probe': probe,
                    'frac_of_sustained': round(kr[dom]['tflops'] / sustained, 4),
                    'mfma_issue_frac_of_sustained': round(per_alg * kr[dom]['tflops'] / sustained, 4),
                    'mfma_executed_frac_of_sustained': round(executed / sustained, 4),
                    'sustained_note': 'frac_of_sustained = algorithmic (direct-convolution) flops / sustained; mfma_issue = x MFMA flops issued '
                                      'per algorithmic flop (K9: 3 = three f16 MFMAs per fp32-grade product; K17: 3 x 16/36, Winograd F(2x2,3x3) '
                                      'needs 16 products per 2x2 outputs instead of 36); mfma_executed also counts the MFMAs spent on channel / '
                                      'tile padding (what the pipe actually ran)',
                    # HBM/fabric bytes per launch from the committed PMC passes (null when they do not match this geometry
                    # and precision); `committed_profile` names the file / commit they were measured at
                    'traffic': (tr or {}).get('bytes'), 'committed_profile': tr,
                    'launch_ms': round(kr[dom]['ms'], 3), 'launch_ms_rocprof': rocprof_launch_ms(dom),
                    'mfma_issue_frac': round(kr[dom]['mfma_issue_frac'], 4),
                    'same_launch_on_k9': {'launch_ms': round(kr['k_conv[K9 3x3 196->196 @240x320]']['ms'], 3),
                                          'frac': round(kr['k_conv[K9 3x3 196->196 @240x320]']['frac'], 4)} if wino else None,
                    'note': ('algorithmic convolution flops per launch (2 N H W Cin Cout 9, the direct form) / event-timed launch duration '
                             'against the dense f16 MFMA peak.  K17 executes 1.33 f16 MFMA flops per algorithmic flop (split operands x '
                             'Winograd), so the matrix pipe is busy mfma_issue_frac of the time at the nominal 2.4 GHz; what bounds it is not '
                             'the matrix pipe but the L2 -> LDS operand stream: a workgroup holds 16 accumulator planes for its 256 outputs x 64 '
                             'channels (half the register file of a CU) and re-streams the 16 transformed weight planes for every such tile -- '
                             '26 GB through L2 per launch of the 128-channel layer in 1.85 ms (DESIGN section 4, K17; '
                             'profiles/r04_k17_winograd.txt)') if wino else
                            ('algorithmic convolution flops per launch / event-timed launch duration against the dense f16 '
                             'MFMA peak; the kernel executes 3 f16 MFMAs per fp32-grade product (split operands), so the '
                             'matrix pipe is busy mfma_issue_frac of the time AT THE NOMINAL 2.4 GHz; the shader clock '
                             'measured inside this kernel is 1.5-1.8 GHz (power limit under dense MFMA: s_memtime stamps in '
                             'profiles/r01_k9_workgroup_timeline.txt, GRBM_GUI_ACTIVE in profiles/r02_pmc_util.txt: 1.59 GHz, '
                             'SQ_VALU_MFMA_BUSY_CYCLES = 72 % of those cycles)')}
        res = {
            'metric': 'image-pairs/sec (match+solve+regress) at 640x480',
            'value': round(world * a.pairs * a.steps / dt, 3), 'unit': 'image-pairs/sec',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'prime_steps': prime, 'prime_cap': max(0, 3 - a.warmup) + a.warmup + SETTLE_EXTRA,
            'ms_per_step': round(1000 * dt / a.steps, 3), 'per_rank_ms_per_step': per_rank_ms,
            # host time of a step without its waits for the GPU (host_work_ms: Python + ctypes + the caching allocator), measured on
            # three extra untimed steps: what N ranks on one host contend for
            'host_ms_per_step': max(per_rank_host_ms), 'per_rank_host_ms_per_step': per_rank_host_ms,
            # normal hosts: ~5 ms.  A box whose host is an order of magnitude slower (seen once in ~15 runs of round 6: 42.9 ms, step 113.5 ms
            # with unchanged kernels) cannot keep the GPU fed behind the step's match-count read: the line then measures the host
            'host_bound': bool(max(per_rank_host_ms) > 0.2 * 1000 * dt / a.steps),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if a.precision == 'fp32' else f'f32 tensors/accumulation, {a.precision} matrix operands',
            'dtype_note': ('f32 = fp32 tensors, fp32 accumulation; matrix products on the f16 MFMA pipe as split-f16x3 operand '
                           'pairs (hi*hi + hi*lo + lo*hi, 22 significand bits: fp32-grade, not a bitwise fmaf chain); '
                           'solver float64.  Activation range: every split-fp16 launch reports an operand beyond its range (|a| > 4094 at '
                           'the default exponent) through a device flag; LoFTR.forward re-runs at a 16x wider range when it fires '
                           f'(this run: activation exponent {model.act_exp}, 4 = default = no re-run)') if a.precision == 'fp32' else None,
            'data': 'synthetic',
            'config': {'workload': 'Matterport3D-shaped eval (BASELINE configs[1]): batch 32 pairs @ 640x480 per GPU, '
                                   'LoFTR match + 8-pt prior-RANSAC solve (x2 rounds) + EMM head (x2: the head\'s feature stage -- 2 LoFTR '
                                   'layers, CrossAttention K2, CrossBlock MLP -- does not read the solver numbers and is computed ONCE per '
                                   'step, the second call re-runs only the 13-number-dependent MLPs / gate on it; exact reuse, '
                                   'tests/test_pipeline_gpu.py), seeded random weights',
                       'pairs_per_gpu': a.pairs, 'hypotheses': a.hyp, 'minimal_solver': a.minimal, 'fine_pred_steps': 2,
                       'matches_per_pair': round(matches, 1), 'solver_success_frac': ok_frac, 'pose_error': pose_err,
                       'parallelism': f'dp{world} (independent pairs, no data-path collective)'},
            'roofline': roof,
            'kernels': {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in kr.items()},
            'build_id': _build_id(),
            'process_group': {'backend': dist.get_backend(), 'world_size': dist.get_world_size()} if dist is not None else None,
        }
        if world == 1 and a.precision == 'fp32' and not a.no_other_modes:
            # INFORMATIONAL, never `value`: the same step with 16-bit matrix operands in some or all stages (LoFTR.set_precision).  The
            # reference computes in fp32 throughout (no autocast anywhere in it), so the fp32-grade line above is the only parity
            # configuration; every entry below states what it costs: match-set IoU against the parity line (ids are no longer
            # bit-exact => north_star's index clause fails), the solver's pose error on these pairs, the sub-pixel deviation.
            L2 = 4800 * 4800

            def keys_of(out):
                return out['b_ids'].long() * L2 + out['i_ids'].long() * 4800 + out['j_ids'].long()
            k32, order32 = torch.sort(keys_of(last))
            mk32 = last['mkpts1_f'][order32]
            parity_err = pose_errors(last)
            # the yardstick for everything below: the SAME fp32-grade step under other RANSAC sampling seeds.  On these synthetic
            # pairs (small baseline, a random-weight head as the second round's prior) the median solver error moves by as much as
            # the 16-bit modes move it -- which hypothesis wins is sensitive to any perturbation of the match set
            seed_spread = [pose_errors(step(seed=sd)) for sd in (1, 2, 3, 4)]

            def timed_mode(mode, n=5):
                model.set_precision(mode)
                for _ in range(2):
                    step()
                settle(step, 0, dev, host_frac=0.45 if a.pairs >= 8 else None)
                fence()
                t1 = time.perf_counter()
                for _ in range(n):
                    out = step()
                fence()
                dtm_ = (time.perf_counter() - t1) / n
                model.set_precision('fp32')
                k = keys_of(out)
                pos = torch.searchsorted(k32, k).clamp_(max=k32.numel() - 1)
                common = k32[pos] == k
                ncommon = int(common.sum())
                dev_px = (out['mkpts1_f'][common] - mk32[pos[common]]).abs().amax(1) if ncommon else torch.zeros(1, device=dev)
                pe = pose_errors(out)
                acc = dict(pe, matches_per_pair=round(k.numel() / a.pairs, 1),
                           match_set_iou_vs_parity_line=round(ncommon / max(k.numel() + k32.numel() - ncommon, 1), 4),
                           ids_bit_exact=bool(k.numel() == k32.numel() and ncommon == k.numel()),
                           mkpts1_f_dev_px={'median': round(float(dev_px.median()), 5), 'max': round(float(dev_px.max()), 4)},
                           solver_R_err_vs_parity_line=round(pe['solver_median_R_deg'] / max(parity_err['solver_median_R_deg'], 1e-9), 3))
                return dtm_, acc

            def entry(dtm_, acc, note):
                # host_bound: the host work of a step (the same in every mode, measured above) against THIS mode's step time -- a fast mode
                # on a box with a slow host measures the host (seen in round 6: 35 ms of host work, fp16 at 105 ms next to fp32 at 82)
                return {'value': round(a.pairs / dtm_, 3), 'ms_per_step': round(1000 * dtm_, 3),
                        'host_bound': bool(res['host_ms_per_step'] > 0.2 * 1000 * dtm_), 'note': note, 'accuracy': acc}
            (dt16, acc16), (dtm, accm), (dtf, accf) = timed_mode('fp16'), timed_mode('mixed16'), timed_mode('fp16-fine')
            res['other_modes'] = {
                'parity_line_pose_error': parity_err,
                'parity_line_pose_error_other_seeds': {'solver_median_R_deg': [e['solver_median_R_deg'] for e in seed_spread],
                                                       'solver_median_t_deg': [e['solver_median_t_deg'] for e in seed_spread],
                                                       'note': 'the parity line itself under RANSAC seeds 1-4: the noise floor of the solver errors '
                                                               'quoted for the modes below (synthetic pairs; not an accuracy statement)'},
                'fp16_fine': entry(dtf, accf, "plain fp16 operands in the FPN's fine branch only: every coarse decision (ids, mconf) bit-identical "
                                              'to the parity line, sub-pixel refinement deviates ~0.01 px'),
                'fp16_operands': entry(dt16, acc16, '16-bit operands in the large matrix products of the step (plain fp16 in K9 incl. the fused k|v-state / '
                                                    'q-apply launches, K13 / K14, K2; bf16 in K1), fp32 tensors and accumulation: NARROWER than the '
                                                    "reference's fp32 arithmetic -- ids not bit-exact, solver pose error as stated; informational"),
                'mixed16': entry(dtm, accm, "between the two: as fp16_operands, but the encoder layers' attention-state launches and the fine level "
                                            '(K13 / K14) on split-fp16 operands; informational, not the parity line')}
            # one stage at a time in 16-bit: where the pose-error cost of the 16-bit modes comes from
            stages = {}
            for st in (('trunk',), ('fpn',), ('coarse_dense',), ('coarse_dense', 'coarse_state'), ('fine_layers',), ('k1',), ('k2',)):
                dts, accs = timed_mode(st, n=3)
                stages['+'.join(st)] = entry(dts, accs, 'only this stage on 16-bit operands')
            ok = [name for name, e in stages.items() if e['accuracy']['solver_R_err_vs_parity_line'] <= 1.1]
            union = tuple(sorted({x for name in ok for x in name.split('+')}, key=model.STAGES.index))
            if union:
                dtu, accu = timed_mode(union, n=3)
                stages['union_of_stages_within_1.1x'] = entry(dtu, accu, 'stages: ' + ', '.join(union) + ' -- every stage whose own solver rotation error '
                                                                         'stays within 1.1x of the parity line, together')
            res['other_modes']['precision_stages'] = stages
        if world == 1 and not a.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(a.cpu_pairs, a.hyp)
        if per_rank_peak is not None:
            res['per_rank_sustained_peak'] = per_rank_peak
        if world == 1 and a.precision == 'fp32' and a.pairs == PAIRS_PER_GPU and not a.no_other_workloads:
            res['other_workloads'] = other_workloads(a)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
