"""Few-row Linear launches at the training step's shapes, for `rocprofv3 --kernel-trace --stats -- python3 tools/lin_small_time.py`
(kernel durations by grid size come from the trace; host-side event timing would measure the interpreter)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from far_amd import ops  # noqa: E402

if os.environ.get('FAR_LS_K9'):
    from far_amd import _lib
    _lib.load().far_set_tuning(7, 1)           # comparison: K9 on full-height tiles
g = torch.Generator(device='cuda').manual_seed(0)
for rows, K, N in [(9600, 256, 256), (19200, 256, 256), (19200, 512, 512), (19200, 512, 256), (7500, 128, 128), (4800, 256, 256)]:
    x = torch.randn(rows, K, device='cuda', generator=g)
    pc = ops.PackedConv(torch.randn(N, K, device='cuda', generator=g) * 0.05)
    for _ in range(20):
        ops.linear_f16s(x, pc)
    torch.cuda.synchronize()
