"""Development aid: race hunt for K9.  Its explicit memory waits count requests (conv_igemm_f16s.hip, "memory waits");
an under-wait would show as run-to-run differences.  Every shape is run many times under load and every output must
be bit-identical to the first run (and close to the float64 convolution)."""
import sys
sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
from far_amd import ops

dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(11)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
shapes = [  # N, H, W, Cin, Cout, ks, stride
    (8, 120, 160, 196, 196, 3, 1), (8, 120, 160, 128, 128, 3, 1), (8, 60, 80, 256, 256, 3, 1), (8, 120, 160, 128, 196, 3, 2),
    (4, 97, 131, 196, 128, 3, 1), (1, 1, 153600, 256, 256, 1, 1), (1, 1, 153600, 512, 256, 1, 1), (1, 1, 300000, 128, 384, 1, 1),
    (8, 120, 160, 196, 256, 1, 1), (3, 37, 45, 100, 300, 3, 1), (2, 64, 64, 36, 64, 3, 1),
]
bad = 0
for (N, H, W, Cin, Cout, ks, st) in shapes:
    for split in (True, False):
        x = torch.randn(N, H, W, Cin, device=dev, generator=g)
        w = torch.randn(Cout, Cin, ks, ks, device=dev, generator=g) * (1.5 / (Cin * ks * ks)) ** 0.5
        pc = ops.PackedConv(w, None, None, split=split, stride=st)
        kw = {}
        if ks == 1 and Cout in (128, 256) and N == 1:
            kw = dict(ln=(torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev), 1e-5))
        first = ops.conv_nhwc(x, pc, act='relu' if not kw else 'none', **kw)
        if not kw and N * H * W * Cin * Cout * ks * ks < 3e11:
            ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), stride=st, padding=ks // 2)).permute(0, 2, 3, 1)
            err = float((first.double() - ref).abs().max() / ref.abs().max())
            assert err < (5e-6 if split else 5e-3), (N, H, W, Cin, Cout, ks, st, split, err)
        # other work in flight on a second stream so that timing varies from run to run
        side = torch.cuda.Stream()
        junk = torch.randn(4096, 4096, device=dev)
        ndiff = 0
        for r in range(reps):
            with torch.cuda.stream(side):
                for _ in range(r % 3):
                    junk = junk @ junk * 1e-3
            y = ops.conv_nhwc(x, pc, act='relu' if not kw else 'none', **kw)
            if not torch.equal(y, first):
                ndiff += 1
        torch.cuda.synchronize()
        print(f'{(N, H, W, Cin, Cout, ks, st)} split={split} ln={bool(kw)}: {ndiff} of {reps} runs differ')
        bad += ndiff
print('TOTAL differing runs:', bad)
sys.exit(1 if bad else 0)
