"""Development aid: event-timed K9 launches at the step's shapes, for A/B runs of two builds of the library.

  python tools/k9_ab.py [path/to/libfar_hip_variant.so]
"""
import sys
sys.path.insert(0, '.')
import torch
from far_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
from far_amd import ops

dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1)
R = 64 * 4800


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def lin(cin, cout):
    return ops.PackedConv(torch.randn(cout, cin, device=dev, generator=g) * (1.0 / cin) ** 0.5)


x = torch.randn(1, 1, R, 256, device=dev, generator=g)
m = torch.randn(1, 1, R, 256, device=dev, generator=g)
h512 = torch.randn(1, 1, R, 512, device=dev, generator=g)
gm, bt = torch.ones(256, device=dev), torch.zeros(256, device=dev)
p256, pqkv, pm0, pm2 = lin(256, 256), lin(256, 768), lin(512, 512), lin(512, 256)
out = torch.empty_like(x)
rows = []
rows.append(('linear 256->256', timed(lambda: ops.conv_nhwc(x, p256, out=out))))
rows.append(('qkv 256->768 planes', timed(lambda: ops.conv_nhwc(x, pqkv, out_planes=3))))
rows.append(('merge 256->256 + LN', timed(lambda: ops.conv_nhwc(x, p256, ln=(gm, bt, 1e-5), out=out))))
rows.append(('mlp0 cat[x,msg] 512->512 relu', timed(lambda: ops.conv_nhwc(x, pm0, x2=m, act='relu'))))
rows.append(('mlp2 512->256 + LN + res', timed(lambda: ops.conv_nhwc(h512, pm2, ln=(gm, bt, 1e-5), post_residual=x, out=out))))
del h512, m
Rf = 60000 * 25
xf = torch.randn(1, 1, Rf, 128, device=dev, generator=g)
p128, p384, p256f = lin(128, 128), lin(128, 384), lin(128, 256)
rows.append(('fine 128->128 (x3 = q, k, v)', 3 * timed(lambda: ops.conv_nhwc(xf, p128))))
rows.append(('fine 128->384 planes (fused qkv)', timed(lambda: ops.conv_nhwc(xf, p384, out_planes=3))))
rows.append(('fine 128->256 planes (fused kv)', timed(lambda: ops.conv_nhwc(xf, p256f, out_planes=2))))
del xf
for C, H, W in ((128, 240, 320), (196, 120, 160), (256, 60, 80)):
    xi = torch.randn(64, H, W, C, device=dev, generator=g).relu_()
    pc = ops.PackedConv(torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.03, torch.ones(C, device=dev), torch.zeros(C, device=dev))
    rows.append((f'3x3 {C}->{C} @{H}x{W}', timed(lambda: ops.conv_nhwc(xi, pc, act='relu'), 10)))
    rows.append((f'3x3 {C}->{C} @{H}x{W} + res', timed(lambda: ops.conv_nhwc(xi, pc, act='relu', residual=xi), 10)))
    del xi
for Ci, Co, H, W in ((128, 196, 240, 320), (196, 256, 120, 160)):
    xi = torch.randn(64, H, W, Ci, device=dev, generator=g)
    up = torch.randn(64, H // 2, W // 2, Co, device=dev, generator=g)
    pc = ops.PackedConv(torch.randn(Co, Ci, 1, 1, device=dev, generator=g) * 0.05)
    rows.append((f'FPN merge {Ci}->{Co} @{H}x{W}: conv, then K8', timed(lambda: ops.upsample2x_add(up.permute(0, 3, 1, 2), ops.conv_nhwc(xi, pc).permute(0, 3, 1, 2)), 10)))
    rows.append((f'FPN merge {Ci}->{Co} @{H}x{W}: fused', timed(lambda: ops.conv_nhwc(xi, pc, up=up), 10)))
    del xi, up
print(f'# {_lib.LIB_PATH}')
for nm, us in rows:
    print(f'{nm:48s} {us:9.1f} us')
