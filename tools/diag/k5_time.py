"""Diagnostic: K5 forward timings at the bench shapes (coarse: 32 maps x 4800 x 256; fine: 61k windows x 25 x 128)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far_amd import _lib, ops          # noqa: E402


def t_ms(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


lib = _lib.load()
for (N, L, C) in ((32, 4800, 256), (61000, 25, 128)):
    q, k, v = (torch.randn(N, L, C, device='cuda') for _ in range(3))
    for knob in (0, 1):
        lib.far_set_tuning(4, knob)
        ms = t_ms(lambda: ops.linear_attention(q, k, v, 8))
        print(f'N={N} L={L} C={C} knob4={knob}: {ms * 1e3:8.1f} us  ({4 * N * L * C * 4 / ms / 1e9:.2f} TB/s on q,k,v,out)')
    lib.far_set_tuning(4, 0)

q, k, v = (torch.randn(32, 4800, 256, device='cuda') for _ in range(3))
for tpu in (2, 3, 4, 6, 8, 16, 32):
    lib.far_set_tuning(5, tpu)
    ms = t_ms(lambda: ops.linear_attention(q, k, v, 8))
    print(f'coarse tiles_per_unit={tpu}: {ms * 1e3:8.1f} us')
lib.far_set_tuning(5, 0)

for chunk in (96, 160, 224, 320, 480, 640):
    lib.far_set_tuning(6, chunk)
    ms = t_ms(lambda: ops.linear_attention(q, k, v, 8))
    print(f'coarse fwd chunk={chunk}: {ms * 1e3:8.1f} us')
lib.far_set_tuning(6, 0)
