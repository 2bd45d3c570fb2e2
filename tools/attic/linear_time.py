"""The five K9 Linear launches of a coarse LoFTR encoder layer at the bench size (32 pairs x 2 images x 4800 tokens = 307 200 rows,
d_model 256) against their floors: HBM bytes at 5.4 TB/s and split-fp16 MFMA flops at the sustained rate.  Usage: python tools/linear_time.py"""
import sys
sys.path.insert(0, '.')
import torch
import bench
from far_amd import ops

R, d = 307200, 256
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(R, d, device='cuda', generator=g)
m = torch.randn(R, d, device='cuda', generator=g)
hid = torch.randn(R, 2 * d, device='cuda', generator=g).relu_()
W = lambda o, i: torch.randn(o, i, device='cuda', generator=g) * (1.0 / i) ** 0.5
ln = (torch.ones(d, device='cuda'), torch.zeros(d, device='cuda'), 1e-5)
cases = {
    'q     256->256': (lambda pc: ops.linear_f16s(x, pc), W(d, d), 2 * R * d * 4, 2.0 * R * d * d),
    'k|v   256->512 (2 planes)': (lambda pc: ops.linear_f16s(x, pc, out_planes=2), W(2 * d, d), 3 * R * d * 4, 2.0 * R * d * 2 * d),
    'merge 256->256 + LN': (lambda pc: ops.linear_f16s(m, pc, ln=ln), W(d, d), 2 * R * d * 4, 2.0 * R * d * d),
    'mlp0  [x|m] 512->512 + ReLU': (lambda pc: ops.linear_f16s(x, pc, act='relu', x2=m), W(2 * d, 2 * d), 4 * R * d * 4, 2.0 * R * 4 * d * d),
    'mlp2  512->256 + LN + x': (lambda pc: ops.linear_f16s(hid, pc, ln=ln, post_residual=x), W(d, 2 * d), 4 * R * d * 4, 2.0 * R * 2 * d * d),
}
tot = [0.0, 0.0, 0.0]
for name, (fn, w, nbytes, flops) in cases.items():
    pc = ops.PackedConv(w)
    t = min(bench.event_time_ms(lambda: fn(pc), iters=10, warm=3) for _ in range(3))
    hbm, mf = nbytes / 5.4e12 * 1e3, 3 * flops / 1.7e15 * 1e3
    tot[0] += t; tot[1] += hbm; tot[2] += mf
    print(f'{name:32s} {1e3 * t:6.0f} us   HBM floor {1e3 * hbm:5.0f} us   MFMA floor {1e3 * mf:5.0f} us   -> {t / max(hbm, mf):.2f}x the larger floor')
print(f'layer: {1e3 * tot[0]:.0f} us measured, {1e3 * tot[1]:.0f} us HBM, {1e3 * tot[2]:.0f} us MFMA')

# the attention core around them: unfused (k | v stored, K5's three launches) against the k | v projection that ends in K'^T V
N, S = R // 4800, 4800
wkv = W(2 * d, d)
pkv2 = ops.PackedConv(wkv)
pst = ops.PackedConv(ops.kv_interleaved_weight(wkv[:d], wkv[d:], 8))
q = torch.randn(N, S, d, device='cuda', generator=g)
xs = x.view(N, S, d)
def unfused():
    k, v = ops.linear_f16s(xs, pkv2, out_planes=2)
    return ops.linear_attention(q, k, v, 8)
def fused():
    return ops.linear_attention_apply(q, ops.linear_kv_state(xs, pst, S), 8, S)
pq = ops.PackedConv(W(d, d))
_, image = ops.linear_kv_state(xs, pst, S, want_image=True)
def unfused_all():
    qq = ops.linear_f16s(xs, pq)
    k, v = ops.linear_f16s(xs, pkv2, out_planes=2)
    return ops.linear_attention(qq, k, v, 8)
def fused_all():
    _, im = ops.linear_kv_state(xs, pst, S, want_image=True)
    return ops.linear_q_apply(xs, pq, im, S)
for name, fn in (('k|v + K5 (three launches)', unfused), ('k|v -> K^T V state + apply', fused), ('   the state launch alone', lambda: ops.linear_kv_state(xs, pst, S)),
                 ('q, k|v, K5: five launches', unfused_all), ('state + q-with-apply: 3 launches', fused_all),
                 ('   q-with-apply alone', lambda: ops.linear_q_apply(xs, pq, image, S))):
    t = min(bench.event_time_ms(fn, iters=10, warm=3) for _ in range(3))
    print(f'{name:32s} {1e3 * t:6.0f} us')
