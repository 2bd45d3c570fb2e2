import sys, torch
sys.path.insert(0, '/root/repo')
from far_amd import ops
g = torch.Generator(device='cuda').manual_seed(1)
img = torch.rand(64, 1, 480, 640, device='cuda', generator=g)
w = torch.randn(128, 1, 7, 7, device='cuda', generator=g) * 0.1
sc = torch.rand(128, device='cuda', generator=g) + 0.5; sh = torch.randn(128, device='cuda', generator=g) * 0.1
y0 = ops.stem7x7(img, w, sc, sh)
for _ in range(5): ops.stem7x7(img, w, sc, sh)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for rep in range(5):
    e0.record()
    for _ in range(20): y = ops.stem7x7(img, w, sc, sh)
    e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
print('stem ms', ' '.join(f'{t:.4f}' for t in ts), 'checksum', float(y.double().sum()), 'equal to first', bool(torch.equal(y, y0)))
