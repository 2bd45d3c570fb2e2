"""A/B of K9's seven-tile mode on the 196-output-channel 3x3 layers (far_set_tuning key 4: 1 = eight-tile kernel)."""
import sys
sys.path.insert(0, '.')
import torch
import bench
from far_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device='cuda').manual_seed(1)
shapes = {'196->196 @240x320': (240, 320, 196, 196), '196->196 @120x160': (120, 160, 196, 196), '256->196 @120x160': (120, 160, 256, 196)}
for label, (H, W, ci, co) in shapes.items():
    x = torch.randn(64, H, W, ci, device='cuda', generator=g).relu_()
    pc = ops.PackedConv(torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (ci * 9)) ** 0.5, torch.ones(co, device='cuda'), torch.zeros(co, device='cuda'))
    res = {}
    outs = {}
    for rnd in range(3):
        for mode in (1, 0):
            lib.far_set_tuning(4, mode)
            t = bench.event_time_ms(lambda: ops.conv_nhwc(x, pc, act='relu'), iters=5, warm=2)
            res.setdefault(mode, []).append(t)
            outs[mode] = ops.conv_nhwc(x, pc, act='relu')
    lib.far_set_tuning(4, 0)
    fl = 2.0 * 64 * H * W * ci * co * 9
    print(f'{label}: eight tiles {min(res[1]):.3f} ms  seven tiles {min(res[0]):.3f} ms  ({100 * (min(res[0]) / min(res[1]) - 1):+.1f} %)  '
          f'-> {fl / min(res[0]) / 1e9:.1f} TFLOP/s = {fl / min(res[0]) / 1e9 / 2500:.4f} of peak; identical outputs: {bool(torch.equal(outs[0], outs[1]))}')
