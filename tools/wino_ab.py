"""K17 (Winograd F(2x2,3x3), split fp16) against K9 and a float64 convolution: errors on small shapes, then same-box timings
at the bench shapes (64 images).  Usage: python tools/wino_ab.py [--quick] [--no-time]"""
import os
import sys
sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
import bench
from far_amd import _lib, ops

lib = _lib.load()
quick = '--quick' in sys.argv


def rel(a, ref):
    d = (a.double() - ref).abs()
    return float(d.max() / ref.abs().max()), float((d.pow(2).mean() / ref.pow(2).mean()).sqrt())


def check(N, H, W, ci, co, seed=0, mix=0):
    g = torch.Generator(device='cuda').manual_seed(seed + ci + co + H)
    x = (torch.randn(N, H, W, ci, device='cuda', generator=g) * 1.5).relu_()
    w = torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (ci * 9)) ** 0.5
    scale = torch.rand(co, device='cuda', generator=g) + 0.5
    shift = torch.randn(co, device='cuda', generator=g) * 0.1
    res = torch.randn(N, H, W, co, device='cuda', generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1) * scale.double() + shift.double()
    lib.far_set_tuning(8, mix)
    pw = ops.PackedWino(w, scale, shift)
    pc = ops.PackedConv(w, scale, shift)
    out = []
    for act, r in (('none', None), ('relu', res), ('leaky', None)):
        y = ops.conv3x3_wino(x, pw, residual=r, act=act, slope=0.01)
        y9 = ops.conv_nhwc(x, pc, residual=r, act=act, slope=0.01)
        rr = ref + (r.double() if r is not None else 0)
        rr = {'none': lambda t: t, 'relu': torch.relu, 'leaky': lambda t: F.leaky_relu(t, 0.01)}[act](rr)
        out.append((act, rel(y, rr), rel(y9, rr)))
    y2 = ops.conv3x3_wino(x, pw, residual=res, act='relu')
    y3 = ops.conv3x3_wino(x, pw, residual=res, act='relu')
    det = bool(torch.equal(y2, y3))
    torch.cuda.synchronize()
    print(f'N{N} {H}x{W} {ci}->{co} mix={mix}: ' + '  '.join(f'{a}: wino {e[0]:.2e}/{e[1]:.2e} k9 {k[0]:.2e}/{k[1]:.2e}' for a, e, k in out)
          + f'  deterministic {det}  overflow {bool(ops.overflow_flag("cuda").item())}', flush=True)
    lib.far_set_tuning(8, 0)


for mix in (() if os.environ.get('WINO_NOCHECK') else (1, 0)):
    check(1, 16, 16, 16, 64, mix=mix)
    check(1, 16, 16, 32, 32, mix=mix)
    check(2, 24, 40, 32, 64, mix=mix)
    check(3, 30, 37, 196, 196, mix=mix)
    check(1, 17, 16, 128, 128, mix=mix)
    check(2, 9, 50, 256, 196, mix=mix)
    check(1, 5, 7, 196, 128, mix=mix)
if not quick:
    check(4, 240, 320, 128, 128)
    check(4, 120, 160, 256, 256)

if '--no-time' not in sys.argv:
    g = torch.Generator(device='cuda').manual_seed(1)
    shapes = {'128->128 @240x320': (240, 320, 128, 128), '196->196 @240x320': (240, 320, 196, 196), '196->128 @240x320': (240, 320, 196, 128),
              '256->256 @120x160': (120, 160, 256, 256), '196->196 @120x160': (120, 160, 196, 196), '256->196 @120x160': (120, 160, 256, 196),
              '256->256 @60x80': (60, 80, 256, 256)}
    if quick:
        shapes = {'128->128 @240x320': (240, 320, 128, 128)}
    for label, (H, W, ci, co) in shapes.items():
        NB = int(os.environ.get('WINO_N', '64'))
        x = torch.randn(NB, H, W, ci, device='cuda', generator=g).relu_()
        w = torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (ci * 9)) ** 0.5
        pc = ops.PackedConv(w, torch.ones(co, device='cuda'), torch.zeros(co, device='cuda'))
        pw = ops.PackedWino(w, torch.ones(co, device='cuda'), torch.zeros(co, device='cuda'))
        tk, tw, tw5 = [], [], []
        for rnd in range(3):
            tk.append(bench.event_time_ms(lambda: ops.conv_nhwc(x, pc, act='relu'), iters=5, warm=2))
            lib.far_set_tuning(8, 0)
            tw.append(bench.event_time_ms(lambda: ops.conv3x3_wino(x, pw, act='relu'), iters=5, warm=2))
            lib.far_set_tuning(8, 1)
            tw5.append(bench.event_time_ms(lambda: ops.conv3x3_wino(x, pw, act='relu'), iters=5, warm=2))
            lib.far_set_tuning(8, 0)
        fl = 2.0 * NB * H * W * ci * co * 9
        print(f'{label} x{NB}: K9 {min(tk):.3f} ms  K17 {min(tw):.3f} ms (split2: {min(tw5):.3f})  speedup {min(tk) / min(tw):.2f}x  '
              f'-> {fl / min(tw) / 1e9:.1f} TFLOP/s direct-equivalent = {fl / min(tw) / 1e9 / 2500:.4f} of peak', flush=True)
