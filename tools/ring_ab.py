#!/usr/bin/env python
"""Round 6, the LDS-DMA ring race on the REAL kernels (the ring-only reproducer is tools/ubench/ring_war.hip).

  python tools/ring_ab.py --build     (CPU) builds far_amd/lib/exp/libfar_ringexp.so: the product objects, with attn_block_f16s.hip
                                      and mlp_fused_f16s.hip recompiled under -DFAR_RING_EXP (far_set_tuning(11, v): v = 0 / 1 / 2 are the
                                      product's K14 pipelines -- <4 waves, 3 slots, counted> and <4, 3, vmcnt(0)> run TWO workgroups
                                      per CU, <8, 4, counted> one -- and v = 4 / 5 / 6 the same three with the barrier WITHOUT its
                                      lgkmcnt(0), i.e. the rounds-3..5 code; K13: v & 4 likewise)
  FAR_HIP_LIB=far_amd/lib/exp/libfar_ringexp.so python tools/ring_ab.py [--windows 60148] [--launches 20]
                                      (GPU) every variant: event-timed, then `launches` launches next to a busy second stream, windows
                                      that differ from the first launch and from the shipped form counted.
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXP_LIB = os.path.join(ROOT, 'far_amd', 'lib', 'exp', 'libfar_ringexp.so')


def build():
    from far_amd import build as B
    B.build(verbose=False)
    os.makedirs(os.path.dirname(EXP_LIB), exist_ok=True)
    objs = []
    procs = []
    for src in B.sources():
        base = os.path.basename(src)
        obj = os.path.join(B.LIBDIR, base[:-4] + '.o')
        if base in ('attn_block_f16s.hip', 'mlp_fused_f16s.hip'):
            obj = os.path.join(os.path.dirname(EXP_LIB), base[:-4] + '.exp.o')
            procs.append(subprocess.Popen([B.HIPCC] + B.FLAGS + ['-DFAR_RING_EXP', '-c', src, '-o', obj]))
        objs.append(obj)
    for p in procs:
        if p.wait() != 0:
            sys.exit('hipcc failed')
    subprocess.check_call([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', EXP_LIB] + objs)
    print(EXP_LIB)


def ev(torch, fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build', action='store_true')
    ap.add_argument('--windows', type=int, default=60148)
    ap.add_argument('--launches', type=int, default=20)
    a = ap.parse_args()
    if a.build:
        return build()
    import torch
    from far_amd import _lib, ops
    lib = _lib.load()
    D, H = 128, 8
    g = torch.Generator(device='cuda').manual_seed(5)
    ws = [torch.randn(D, D, device='cuda', generator=g) / 11 for _ in range(4)]
    gam, bet = torch.rand(D, device='cuda', generator=g) + 0.5, torch.randn(D, device='cuda', generator=g) * 0.1
    w0 = torch.randn(2 * D, 2 * D, device='cuda', generator=g) / 16
    w2 = torch.randn(D, 2 * D, device='cuda', generator=g) / 16
    pa, pm = ops.PackedAttn(*ws), ops.PackedMlp(w0, w2)
    side = torch.cuda.Stream()
    big = torch.randn(64 << 20, device='cuda')                  # 256 MB: the second stream's HBM traffic
    n = 2 * a.windows
    x = torch.randn(n, 25, D, device='cuda', generator=g)
    s = torch.randn(n, 25, D, device='cuda', generator=g)
    # far_set_tuning(11, v) as attn_block_launch reads it: v & 3 selects the pipeline; FAR_RING_EXP builds: v = 4 / 5 / 6 = pipeline 0 / 1 / 2
    # with the barrier WITHOUT its lgkmcnt(0)
    names = {0: '<4 waves, 3 slots, counted>, 2 wg/CU (default)', 1: '<8 waves, 4 slots, counted>, 1 wg/CU (round 5)', 2: '<4, 3, vmcnt(0)>, 2 wg/CU'}
    for kern in ('K14', 'K13'):
        fn = (lambda: ops.attn_block(x, s, pa, H, gam, bet, 1e-5)) if kern == 'K14' else (lambda: ops.mlp_fused(x, s, pm, gam, bet, 1e-5))
        lib.far_set_tuning(11, 0)
        ref = fn()
        for v in ((1, 0, 4, 1, 5, 2, 6, 0) if kern == 'K14' else (0, 4, 0)):      # (the first entry warms the clocks up)
            lib.far_set_tuning(11, v)
            t = ev(torch, fn)
            first = fn()
            bad_runs, bad_windows = 0, 0
            for i in range(a.launches):
                with torch.cuda.stream(side):                   # a busy neighbour: copies + a reduction over 256 MB
                    big2 = big * 1.0001
                    _ = big2.sum()
                y = fn()
                d = int(((y - first).abs().flatten(1).max(1).values > 0).sum())
                bad_runs += d > 0
                bad_windows += d
            torch.cuda.synchronize()
            vs_ref = int(((first - ref).abs().flatten(1).max(1).values > 0).sum())
            what = (names[v & 3] if kern == 'K14' else '2 wg/CU') + (', barrier WITHOUT lgkmcnt(0) (rounds 3-5)' if v & 4 else ', lgkmcnt(0) + barrier (round 6)')
            print(f'{kern} windows {n} variant {v} [{what}]: {t:.3f} ms; launches differing from the first {bad_runs} of {a.launches} '
                  f'({bad_windows} windows in all); windows differing from the shipped form {vs_ref}', flush=True)
    lib.far_set_tuning(11, 0)


if __name__ == '__main__':
    main()
