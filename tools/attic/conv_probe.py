"""GPU probe for K9 (far_conv_nhwc_f32): error against a float64 convolution next to the vendor fp32 kernel's error,
and timings of both.  usage: python tools/conv_probe.py [batch]"""
import sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from far_amd import ops


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(N, H, W, Cin, Cout, ks, split=True, check=True):
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, Cin, H, W, device='cuda', generator=g).relu_() * 1.3
    w = torch.randn(Cout, Cin, ks, ks, device='cuda', generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5
    scale = torch.rand(Cout, device='cuda', generator=g) + 0.5
    shift = torch.randn(Cout, device='cuda', generator=g) * 0.1
    xn = x.permute(0, 2, 3, 1).contiguous()
    res = torch.randn(N, H, W, Cout, device='cuda', generator=g)
    pc = ops.PackedConv(w, scale, shift, split=split)
    y = ops.conv_nhwc(xn, pc, residual=res, act='relu')
    msg = f'N={N} {H}x{W} {Cin}->{Cout} k{ks} split={split}:'
    if check:
        nb = min(N, 2)
        ref = F.conv2d(x[:nb].double(), w.double(), padding=ks // 2)
        ref = torch.relu(ref * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
                         + res[:nb].permute(0, 3, 1, 2).double())
        ven = F.conv2d(x[:nb], w, padding=ks // 2)
        ven = torch.relu(ven * scale[None, :, None, None] + shift[None, :, None, None] + res[:nb].permute(0, 3, 1, 2))
        mine = y[:nb].permute(0, 3, 1, 2).double()
        den = ref.abs().max()
        msg += f' max|err|/max|ref| mine {float((mine - ref).abs().max() / den):.2e} vendor {float((ven.double() - ref).abs().max() / den):.2e}'
        msg += f' rms mine {float((mine - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e} vendor {float((ven.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}'
    t_mine = timeit(lambda: ops.conv_nhwc(xn, pc, residual=res, act='relu'))
    t_ven = timeit(lambda: F.conv2d(x, w, padding=ks // 2))
    fl = 2.0 * N * H * W * Cin * Cout * ks * ks
    msg += f' | mine {t_mine:.3f} ms ({fl / t_mine / 1e9:.0f} TF/s)  vendor conv only {t_ven:.3f} ms ({fl / t_ven / 1e9:.0f} TF/s)'
    print(msg, flush=True)


if __name__ == '__main__':
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    case(2, 24, 40, 32, 64, 3)
    case(2, 30, 37, 196, 196, 3)
    case(1, 1, 1000, 256, 256, 1)
    for split in (True, False):
        case(nb, 240, 320, 128, 128, 3, split)
        case(nb, 240, 320, 196, 196, 3, split)
        case(nb, 240, 320, 196, 128, 3, split)
        case(nb, 120, 160, 196, 196, 3, split)
        case(nb, 120, 160, 256, 256, 3, split)
        case(nb, 60, 80, 256, 256, 3, split)
        case(nb, 240, 320, 128, 196, 1, split)
        case(1, 1, nb * 4800, 256, 256, 1, split)
        case(1, 1, nb * 4800, 512, 512, 1, split)
