"""K9's plain Linear launch and its k | v-state launch (EPI 1: la_kv_pass) ten times each, outputs compared bit for bit -- the
check behind docs/rounds/r06.md section 2g (the half-register write of the fp16 split one slot in front of an MFMA).  Run it on a build
variant with FAR_HIP_LIB=...; on the shipped library both lines must say 0.
python tools/k9_kv_state_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from far_amd import ops
from far_amd.loftr.transformer import LoFTREncoderLayer
torch.manual_seed(3)
layer = LoFTREncoderLayer(256, 8).cuda().eval()
g = torch.Generator(device='cuda').manual_seed(11)
src = torch.randn(8, 4800, 256, device='cuda', generator=g)
with torch.no_grad():
    layer(src, src)
    st = layer._packs._store
    print('packs:', [k for k in st.keys()])
    def get(name):
        v = [v for k, v in st.items() if (k[0] if isinstance(k, tuple) else k) == name][0]
        return v[1] if isinstance(v, tuple) else v
    pq = get('q')
    for tag, fn in (('plain linear q (256->256, EPI 0)', lambda: ops.linear_f16s(src, pq)),
                    ('kv state (EPI 1)', lambda: ops.linear_kv_state(src, get('kv-state'), 4800)),):
        r0 = fn(); r0 = (r0[0] if isinstance(r0, tuple) else r0).clone()
        bad = 0
        for it in range(10):
            r = fn(); r = r[0] if isinstance(r, tuple) else r
            bad += not torch.equal(r, r0)
        print(tag, ':', bad, 'of 10 launches differ')
