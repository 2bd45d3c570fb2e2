"""Development aid: K9 with plain fp16 operands on the backbone's stride-1 3x3 shapes (64 images), event-timed, under
far_set_tuning(11, v) for v in argv (default 0): results must be bit-identical across v.
  [FAR_HIP_LIB=variant.so] python tools/k9_plain_time.py [0 1 2]"""
import sys
sys.path.insert(0, '.')
import torch
from far_amd import _lib, ops

keys = [int(v) for v in sys.argv[1:]] or [0]
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1)
lib = _lib.load()


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rows = {k: [] for k in keys}
for Ci, Co, H, W in ((128, 128, 240, 320), (208, 208, 240, 320), (208, 128, 240, 320), (208, 208, 120, 160), (256, 256, 120, 160),
                     (256, 208, 120, 160), (256, 256, 60, 80)):
    xi = torch.randn(64, H, W, Ci, device=dev, generator=g).relu_()
    pc = ops.PackedConv(torch.randn(Co, Ci, 3, 3, device=dev, generator=g) * 0.03, torch.ones(Co, device=dev), torch.zeros(Co, device=dev),
                        split=False)
    res = xi if Ci == Co else None
    y0 = None
    for k in keys:
        lib.far_set_tuning(11, k)
        y = ops.conv_nhwc(xi, pc, act='relu', residual=res).clone()
        t = timed(lambda: ops.conv_nhwc(xi, pc, act='relu', residual=res))
        same = True if y0 is None else torch.equal(y, y0)
        y0 = y if y0 is None else y0
        rows[k].append(f'{Ci}->{Co}@{H}: {t:.3f} ms {2.0 * 64 * H * W * Ci * Co * 9 / t / 1e9:.0f} TF/s{"" if same else " DIFFERENT BITS"}')
    lib.far_set_tuning(11, 0)
    del xi, y0
for k in keys:
    print(f'tuning(11)={k}: ' + ' | '.join(rows[k]))
