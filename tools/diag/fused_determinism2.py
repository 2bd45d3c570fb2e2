import os, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = sys.argv[1]
shutil.copy(os.path.join(ROOT, 'far_amd', 'lib', f'libfar_hip_m{m}.so'), os.path.join(ROOT, 'far_amd', 'lib', 'libfar_hip.so'))
sys.path.insert(0, ROOT)
import torch
from far_amd import ops
torch.manual_seed(0)
D, H = 128, 8
ws = [torch.randn(D, D, device='cuda') / D ** 0.5 for _ in range(4)]
gam, bet = torch.rand(D, device='cuda') + 0.5, torch.randn(D, device='cuda')
pa = ops.PackedAttn(*ws)
x, s = torch.randn(20000, 25, D, device='cuda'), torch.randn(20000, 25, D, device='cuda')
a0 = ops.attn_block(x, s, pa, H, gam, bet, 1e-5)
bad = sum(int(not torch.equal(ops.attn_block(x, s, pa, H, gam, bet, 1e-5), a0)) for _ in range(30))
print('strict mask', m, ': bad launches', bad, 'of 30')
