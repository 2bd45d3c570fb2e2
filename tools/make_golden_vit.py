"""Container-only: golden G19 -- the 8-Point-ViT shape of K2 (N = 576, h = 3, d = 64, positional index
k*w + j) produced by the reference's own modules:

    interiornetStreetlearn_8ptVit/src/modules/vision_transformer.py
        get_positional_encodings :90-158 (with and without intrinsics), CrossAttention :160-208, CrossBlock :210-234

Weights and inputs come from a seeded CPU generator that tests/test_vit_shape_gpu.py replays (vit_seeded_fill below is the single
definition of it); the fixture holds outputs and the two positional tables only.  Usage: python tools/make_golden_vit.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')


def main():
    from tests.util import vit_seeded_fill, VIT_INTRINSICS
    torch.Tensor.cuda = lambda self, *a, **k: self            # the reference calls .cuda() on the positional table (:190)
    sys.path.insert(0, '/root/reference/interiornetStreetlearn_8ptVit')
    from src.modules import vision_transformer as vt
    blk = vt.CrossBlock(192, 3, qkv_bias=True).eval()
    x = vit_seeded_fill(blk, seed=19)
    intr = torch.tensor([[VIT_INTRINSICS, VIT_INTRINSICS]], dtype=torch.float32)           # (B, 2 frames, [fx, fy, cx, cy])
    with torch.no_grad():
        pos_k = vt.get_positional_encodings(1, 576, intrinsics=intr)[0].numpy()
        pos_0 = vt.get_positional_encodings(1, 576, intrinsics=None)[0].numpy()
        n1, n2 = blk.norm1(x[0:1]), blk.norm1(x[1:2])
        fa, fb = blk.cross_attn(n1, n2, None, intrinsics=intr)
        out = blk(x, intrinsics=intr)
        out0 = blk(x, intrinsics=None)
    path = os.path.join(OUT, 'g19_vit_crossblock.npz')
    np.savez_compressed(path, seed=19, intrinsics=np.asarray(VIT_INTRINSICS, np.float32), pos_intr=pos_k, pos_none=pos_0,
                        xattn_a=fa.numpy(), xattn_b=fb.numpy(), block_out=out.numpy(), block_out_noint=out0.numpy(),
                        note='reference: interiornetStreetlearn_8ptVit vision_transformer.CrossBlock(192, 3, qkv_bias=True), N = 576')
    print(f'g19_vit_crossblock: {os.path.getsize(path) / 1024:.1f} KiB')


if __name__ == '__main__':
    main()
