"""K19 against torch's BatchNorm2d + activation under autograd on the backbone's layer shapes at batch 1 pair (2 images): GPU time per
forward + backward (HIP events over 20 repetitions).  Usage: python tools/bn_time.py"""
import sys
sys.path.insert(0, '.')
import torch
import bench
from far_amd import ops

for (C, H, W) in ((128, 240, 320), (196, 120, 160), (256, 60, 80), (196, 240, 320), (256, 120, 160)):
    x = torch.randn(2, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_()
    up = torch.randn(2, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(C).cuda().train()

    def mine():
        x.grad = None
        ops.bn_act_train(x, bn, 'relu').backward(up)

    def theirs():
        x.grad = None
        torch.relu(bn(x)).backward(up)
    t1 = min(bench.event_time_ms(mine, iters=20, warm=3) for _ in range(3))
    t2 = min(bench.event_time_ms(theirs, iters=20, warm=3) for _ in range(3))
    print(f'C {C} @{H}x{W} x2: K19 {1e3 * t1:.0f} us   torch / MIOpen {1e3 * t2:.0f} us   (forward + backward, wall per call incl. host)')
