"""K18 (F(2,3) along x, split fp16) against K9, K17 and a float64 convolution: errors on small shapes, then same-box timings at the
bench shapes (64 images).  Usage: python tools/wino1d_ab.py [--quick] [--no-time]   (env W1D_NOCHECK=1 skips the error checks)"""
import os
import sys
sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
import bench
from far_amd import _lib, ops

lib = _lib.load()
quick = '--quick' in sys.argv
ops.USE_WINO = False


def rel(a, ref):
    d = (a.double() - ref).abs()
    return float(d.max() / ref.abs().max()), float((d.pow(2).mean() / ref.pow(2).mean()).sqrt())


def check(N, H, W, ci, co, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed + ci + co + H)
    x = (torch.randn(N, H, W, ci, device='cuda', generator=g) * 1.5).relu_()
    w = torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (ci * 9)) ** 0.5
    scale = torch.rand(co, device='cuda', generator=g) + 0.5
    shift = torch.randn(co, device='cuda', generator=g) * 0.1
    res = torch.randn(N, H, W, co, device='cuda', generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1) * scale.double() + shift.double()
    pw = ops.PackedWino1d(w, scale, shift)
    pc = ops.PackedConv(w, scale, shift)
    out = []
    for act, r in (('none', None), ('relu', res), ('leaky', None)):
        y = ops.conv3x3_wino1d(x, pw, residual=r, act=act, slope=0.01)
        y9 = ops.conv_nhwc(x, pc, residual=r, act=act, slope=0.01)
        rr = ref + (r.double() if r is not None else 0)
        rr = {'none': lambda t: t, 'relu': torch.relu, 'leaky': lambda t: F.leaky_relu(t, 0.01)}[act](rr)
        out.append((act, rel(y, rr), rel(y9, rr)))
    y2 = ops.conv3x3_wino1d(x, pw, residual=res, act='relu')
    y3 = ops.conv3x3_wino1d(x, pw, residual=res, act='relu')
    det = bool(torch.equal(y2, y3))
    torch.cuda.synchronize()
    print(f'N{N} {H}x{W} {ci}->{co}: ' + '  '.join(f'{a}: k18 {e[0]:.2e}/{e[1]:.2e} k9 {k[0]:.2e}/{k[1]:.2e}' for a, e, k in out)
          + f'  deterministic {det}  overflow {bool(ops.overflow_flag("cuda").item())}', flush=True)


if not os.environ.get('W1D_NOCHECK'):
    check(1, 16, 32, 16, 64)
    check(1, 16, 32, 32, 32)
    check(2, 24, 40, 32, 64)
    check(3, 30, 37, 196, 196)
    check(1, 17, 16, 128, 128)
    check(2, 9, 50, 256, 196)
    check(1, 5, 7, 196, 128)
    check(1, 40, 70, 208, 208)
    if not quick:
        check(4, 240, 320, 128, 128)
        check(4, 120, 160, 256, 256)

if '--no-time' not in sys.argv:
    g = torch.Generator(device='cuda').manual_seed(1)
    shapes = {'128->128 @240x320': (240, 320, 128, 128), '208->208 @240x320': (240, 320, 208, 208), '208->128 @240x320': (240, 320, 208, 128),
              '256->256 @120x160': (120, 160, 256, 256), '208->208 @120x160': (120, 160, 208, 208), '256->208 @120x160': (120, 160, 256, 208),
              '256->256 @60x80': (60, 80, 256, 256)}
    if quick:
        shapes = {'128->128 @240x320': (240, 320, 128, 128)}
    for label, (H, W, ci, co) in shapes.items():
        NB = int(os.environ.get('WINO_N', '64'))
        x = torch.randn(NB, H, W, ci, device='cuda', generator=g).relu_()
        w = torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (ci * 9)) ** 0.5
        one, zero = torch.ones(co, device='cuda'), torch.zeros(co, device='cuda')
        pc, pw, p1 = ops.PackedConv(w, one, zero), ops.PackedWino(w, one, zero), ops.PackedWino1d(w, one, zero)
        tk, tw, t1 = [], [], []
        for rnd in range(3):
            tk.append(bench.event_time_ms(lambda: ops.conv_nhwc(x, pc, act='relu'), iters=5, warm=2))
            tw.append(bench.event_time_ms(lambda: ops.conv3x3_wino(x, pw, act='relu'), iters=5, warm=2))
            t1.append(bench.event_time_ms(lambda: ops.conv3x3_wino1d(x, p1, act='relu'), iters=5, warm=2))
        print(f'{label} x{NB}: K9 {min(tk):.3f} ms  K17 {min(tw):.3f} ms  K18 {min(t1):.3f} ms  K18 vs K9 {min(tk) / min(t1):.2f}x  vs K17 {min(tw) / min(t1):.2f}x', flush=True)
