#!/usr/bin/env python
"""Round-5 K17 A/Bs: the shipped kernel against far_set_tuning variants on the bench shapes (64 images), same
process, interleaved rounds, minimum of three; outputs compared bit for bit.  Usage: python tools/wino_r05_ab.py KEY=VALUE [KEY=VALUE ...]
(e.g. 15=1: without the raised issue priority of the multiplying wave group; the round's other variants -- 14=1 channel-block walk, 14=2
eight-wave raw requests, 15=3/4 non-temporal requests -- live in commits 7553697 / 6a00f47 / 041fb55, results in profiles/r05_k17_ab.txt)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from far_amd import _lib, ops

lib = _lib.load()
variants = [tuple(int(v) for v in a.split('=')) for a in sys.argv[1:] if '=' in a]
g = torch.Generator(device='cuda').manual_seed(1)
shapes = {'128->128 @240x320': (240, 320, 128, 128), '208->208 @240x320': (240, 320, 208, 208), '208->128 @240x320': (240, 320, 208, 128),
          '256->256 @120x160': (120, 160, 256, 256), '208->208 @120x160': (120, 160, 208, 208), '256->208 @120x160': (120, 160, 256, 208),
          '256->256 @60x80': (60, 80, 256, 256)}
weights = {'128->128 @240x320': 4, '208->208 @240x320': 1, '208->128 @240x320': 1, '256->256 @120x160': 1, '208->208 @120x160': 3,
           '256->208 @120x160': 1, '256->256 @60x80': 3}          # launches of each shape in one step
tot = {None: 0.0}
for label, (H, W, ci, co) in shapes.items():
    x = torch.randn(64, H, W, ci, device='cuda', generator=g).relu_()
    w = torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (ci * 9)) ** 0.5
    pw = ops.PackedWino(w, torch.ones(co, device='cuda'), torch.zeros(co, device='cuda'))
    ref = ops.conv3x3_wino(x, pw, act='relu')
    times = {None: []}
    same = {}
    for rnd in range(3):
        times[None].append(bench.event_time_ms(lambda: ops.conv3x3_wino(x, pw, act='relu'), iters=5, warm=2))
        for k, v in variants:
            lib.far_set_tuning(k, v)
            try:
                y = ops.conv3x3_wino(x, pw, act='relu')
                same[(k, v)] = bool(torch.equal(y, ref))
                times.setdefault((k, v), []).append(bench.event_time_ms(lambda: ops.conv3x3_wino(x, pw, act='relu'), iters=5, warm=2))
            finally:
                lib.far_set_tuning(k, 0)
    line = f'{label} x64: shipped {min(times[None]):.3f} ms'
    tot[None] += weights[label] * min(times[None])
    for kv in variants:
        line += f' | tuning {kv[0]}={kv[1]}: {min(times[kv]):.3f} ms ({min(times[kv]) / min(times[None]):.3f}x, bit-identical {same[kv]})'
        tot[kv] = tot.get(kv, 0.0) + weights[label] * min(times[kv])
    print(line, flush=True)
print('sum over the 14 launches of a step: shipped %.2f ms' % tot[None] + ''.join(f' | tuning {k[0]}={k[1]}: {tot[k]:.2f} ms' for k in variants))
