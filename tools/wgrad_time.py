"""Times K16 against the vendor's backward-weights on the backbone's layer shapes.
python tools/wgrad_time.py  (on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from far_amd import ops  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout, ks, stride   (480 x 640 pair: the stem leaves 240 x 320)
    (2, 240, 320, 128, 128, 3, 1), (2, 240, 320, 196, 196, 3, 1), (2, 240, 320, 196, 128, 3, 1), (2, 240, 320, 128, 196, 3, 2),
    (2, 120, 160, 196, 196, 3, 1), (2, 120, 160, 256, 256, 3, 1), (2, 120, 160, 256, 196, 3, 1), (2, 120, 160, 196, 256, 3, 2),
    (2, 60, 80, 256, 256, 3, 1), (2, 240, 320, 128, 196, 1, 2), (2, 120, 160, 196, 256, 1, 2), (2, 120, 160, 196, 256, 1, 1),
    (1, 300, 32, 256, 256, 1, 1), (1, 300, 32, 256, 512, 1, 1), (1, 300, 32, 512, 256, 1, 1)]


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    tot = {'f16s': 0.0, 'vendor': 0.0}
    for (N, H, W, Cin, Cout, ks, st) in SHAPES:
        x = torch.randn(N, H, W, Cin, device='cuda')
        Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
        dy = torch.randn(N, Ho, Wo, Cout, device='cuda') * 1e-5
        w = torch.zeros(Cout, Cin, ks, ks, device='cuda')
        sc = ops.grad_scale(dy)
        t = {'f16s': timeit(lambda: ops.conv_wgrad(x, dy, ks, st, dy_scale=sc)),
             'vendor': timeit(lambda: torch.ops.aten.convolution_backward(
                 dy.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w, None, [st, st], [ks // 2] * 2, [1, 1], False, [0, 0], 1,
                 [False, True, False]))}
        gf = 2.0 * N * Ho * Wo * Cin * Cout * ks * ks / 1e9
        for k in tot:
            tot[k] += t[k]
        print(f'{N}x{H}x{W} {Cin:3d}->{Cout:3d} k{ks} s{st}: {gf:6.1f} GFLOP  K16 {t["f16s"]:7.1f} us ({gf / t["f16s"] * 1e3:6.1f} TF)  '
              f'vendor {t["vendor"]:7.1f} us')
    print('sum over the list (us):', {k: round(v, 1) for k, v in tot.items()})


if __name__ == '__main__':
    main()
