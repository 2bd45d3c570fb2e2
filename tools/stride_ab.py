"""what a padded pixel stride would be worth for the 196-channel tensors.  The 3x3 kernels pad Cin to 16 per
k-step anyway (196 -> 208), so running the SAME layer on tensors that really have 208 / 224 channels (zero weights in the extra
ones) executes the same matrix work with 832- / 896-byte pixels instead of 784: the time difference is the alignment effect alone.
Prints one JSON object (merged into profiles/rNN_pmc_traffic.json under "padded_stride_experiment").
Usage: python tools/stride_ab.py"""
import json
import sys
sys.path.insert(0, '.')
import torch
import bench
from far_amd import ops

g = torch.Generator(device='cuda').manual_seed(3)
N, H, W = 64, 240, 320
w196 = torch.randn(196, 196, 3, 3, device='cuda', generator=g) * (2.0 / (196 * 9)) ** 0.5
x196 = torch.randn(N, H, W, 196, device='cuda', generator=g).relu_()
out = {'layer': '3x3 196->196 @240x320 x 64 images', 'unit': 'ms per launch (min of 3 rounds of 5)', 'rows': {}}
for cin, cout in ((196, 196), (208, 196), (224, 196), (196, 208), (208, 208), (224, 224), (256, 256)):
    w = torch.zeros(cout, cin, 3, 3, device='cuda')
    w[:196, :196] = w196
    x = torch.zeros(N, H, W, cin, device='cuda')
    x[..., :196] = x196
    one, zero = torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda')
    pc, pw = ops.PackedConv(w, one, zero), ops.PackedWino(w, one, zero)
    ops.USE_WINO = False
    t9 = min(bench.event_time_ms(lambda: ops.conv_nhwc(x, pc, act='relu'), iters=5, warm=2) for _ in range(3))
    ops.USE_WINO = True
    t17 = min(bench.event_time_ms(lambda: ops.conv3x3_wino(x, pw, act='relu'), iters=5, warm=2) for _ in range(3))
    out['rows'][f'Cin {cin} ({4 * cin}-byte input pixels), Cout {cout} ({4 * cout}-byte output pixels)'] = {'K9': round(t9, 3), 'K17': round(t17, 3)}
    print(cin, cout, round(t9, 3), round(t17, 3), file=sys.stderr, flush=True)
    del x, w, pc, pw
print(json.dumps(out))
