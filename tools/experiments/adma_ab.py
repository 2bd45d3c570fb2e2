"""Experiment: K9 Linear launches of a d256 layer under far_set_tuning(11, v): 0 staged (shipped), 1 LDS-DMA activations on 4-wave
tiles, 2 LDS-DMA activations on 8-wave workgroups with a four-slab weight ring.  Times + bit-equality."""
import sys
sys.path.insert(0, '.')
import torch
import bench
from far_amd import ops, _lib
lib = _lib.load()
d = 256
g = torch.Generator(device='cuda').manual_seed(1)
W = lambda o, i: torch.randn(o, i, device='cuda', generator=g) * (1.0 / i) ** 0.5
ln = (torch.ones(d, device='cuda'), torch.zeros(d, device='cuda'), 1e-5)
for R in (153600, 307200):
    x = torch.randn(R, d, device='cuda', generator=g)
    m = torch.randn(R, d, device='cuda', generator=g)
    hid = torch.randn(R, 2 * d, device='cuda', generator=g).relu_()
    N, S = R // 4800, 4800
    wkv = W(2 * d, d)
    pst = ops.PackedConv(ops.kv_interleaved_weight(wkv[:d], wkv[d:], 8))
    pq = ops.PackedConv(W(d, d))
    xs = x.view(N, S, d)
    _, image = ops.linear_kv_state(xs, pst, S, want_image=True)
    cases = {
        'q 256->256': (lambda pc: ops.linear_f16s(x, pc), W(d, d)),
        'merge 256->256 + LN': (lambda pc: ops.linear_f16s(m, pc, ln=ln), W(d, d)),
        'mlp0 512->512 relu': (lambda pc: ops.linear_f16s(x, pc, act='relu', x2=m), W(2 * d, 2 * d)),
        'mlp2 512->256 + LN + x': (lambda pc: ops.linear_f16s(hid, pc, ln=ln, post_residual=x), W(d, 2 * d)),
        'kv state': (lambda pc: ops.linear_kv_state(xs, pst, S, want_image=True), W(d, d)),
        'q apply': (lambda pc: ops.linear_q_apply(xs, pq, image, S), W(d, d)),
    }
    tot = [0, 0, 0]
    for name, (fn, w) in cases.items():
        pc = ops.PackedConv(w)
        row, ref = [], None
        for key in (0, 1, 2):
            lib.far_set_tuning(11, key)
            y = fn(pc)
            y = y[0] if isinstance(y, tuple) else y
            same = True if ref is None else torch.equal(y, ref)
            ref = y.clone() if ref is None else ref
            t = min(bench.event_time_ms(lambda: fn(pc), iters=10, warm=3) for _ in range(3))
            tot[key] += t
            row.append(f'{1e3 * t:6.0f} us{"" if same else " (BITS DIFFER)"}')
        lib.far_set_tuning(11, 0)
        print(f'R={R} {name:26s} staged {row[0]}   dma 4-wave {row[1]}   dma 8-wave {row[2]}')
    print(f'R={R} sum: staged {1e3 * tot[0]:.0f}  dma4 {1e3 * tot[1]:.0f}  dma8 {1e3 * tot[2]:.0f} us')
