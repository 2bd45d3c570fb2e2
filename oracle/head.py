"""Oracle for K2 and the EMM regression head (test infrastructure only, see oracle/__init__.py).

Follows mp3d_loftr/src/loftr/loftr_module/transformer.py: get_positional_encodings :183-248,
CrossAttention.forward :266-303, CrossBlock.forward :335-348,
LocalFeatureTransformerRegressor.forward_emm :423-483.
"""
import numpy as np


def _softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def positional_encodings(h=60, w=80, dtype=np.float32):
    """transformer.py:183-248 with the hard-coded intrinsics of :194-196; returns (h*w, 6).

    The reference evaluates Kinv @ [xs[k], ys[j], 1] per cell in fp32 (:236-240) with
    K = [[fx_n,0,cx_n],[0,fy_n,cy_n],[0,0,1]], fx_n = (517/9)/80*2, fy_n = (517/8)/60*2, cx_n = cy_n = 0,
    so p4 = x / fx_n, p3 = y / fy_n; channels are [p3^2, p4^2, p3*p4, p3, p4, 1] (:243-246), n = j*w + k.
    """
    f32 = np.float32
    fx, fy, cx, cy = f32(517 / 9), f32(517 / 8), f32(40), f32(30)
    hpix, wpix = cy * f32(2), cx * f32(2)
    fxn = (fx / wpix) * f32(2)
    cxn = (cx / wpix) * f32(2) - f32(1)
    fyn = (fy / hpix) * f32(2)
    cyn = (cy / hpix) * f32(2) - f32(1)
    K = np.zeros((3, 3), f32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[2, 2] = fxn, fyn, cxn, cyn, 1
    Kinv = np.linalg.inv(K.astype(np.float64)).astype(f32)   # torch.inverse in fp32; exact here (diagonal)
    import torch
    ys = torch.linspace(-1, 1, steps=h).numpy()                # :198-199 (torch's fp32 linspace, not numpy's)
    xs = torch.linspace(-1, 1, steps=w).numpy()
    p3 = np.empty(h * w, f32)
    p4 = np.empty(h * w, f32)
    for j in range(h):
        for k in range(w):
            vec = Kinv @ np.array([xs[k], ys[j], 1], f32)
            p3[j * w + k] = vec[1] / vec[2]
            p4[j * w + k] = vec[0] / vec[2]
    pos = np.ones((h * w, 6), f32)
    pos[:, 0] = p3 * p3
    pos[:, 1] = p4 * p4
    pos[:, 2] = p3 * p4
    pos[:, 3] = p3
    pos[:, 4] = p4
    return pos.astype(dtype)


def positional_encodings_vit(h=24, w=24, intrinsics=None, dtype=np.float32):
    """interiornetStreetlearn_8ptVit/src/modules/vision_transformer.py:90-158 for one sample; returns (h*w, 6).

    Without intrinsics p3 = ys.repeat(w), p4 = xs.repeat_interleave(h) (:109-110); with intrinsics (fx, fy, cx, cy) the cell loop
    of :146-151 writes index k*w + j (NOT j*w + k as mp3d's table) with Kinv @ [xs[k], ys[j], 1] in fp32."""
    import torch
    f32 = np.float32
    ys = torch.linspace(-1, 1, steps=h).numpy()
    xs = torch.linspace(-1, 1, steps=w).numpy()
    p3 = np.tile(ys, w).astype(f32)
    p4 = np.repeat(xs, h).astype(f32)
    if intrinsics is not None:
        fx, fy, cx, cy = (f32(v) for v in intrinsics)
        hpix, wpix = cy * f32(2), cx * f32(2)
        K = np.zeros((3, 3), f32)
        K[0, 0] = (fx / wpix) * f32(2)
        K[1, 1] = (fy / hpix) * f32(2)
        K[0, 2] = (cx / wpix) * f32(2) - f32(1)
        K[1, 2] = (cy / hpix) * f32(2) - f32(1)
        K[2, 2] = 1
        Kinv = torch.inverse(torch.from_numpy(K)).numpy()
        for j in range(h):
            for k in range(w):
                vec = Kinv @ np.array([xs[k], ys[j], 1], f32)
                p3[k * w + j] = vec[1] / vec[2]
                p4[k * w + j] = vec[0] / vec[2]
    pos = np.ones((h * w, 6), f32)
    pos[:, 0] = p3 * p3
    pos[:, 1] = p4 * p4
    pos[:, 2] = p3 * p4
    pos[:, 3] = p3
    pos[:, 4] = p4
    return pos.astype(dtype)


def bilinear_attention(q, k, vt, scale, dtype=np.float32):
    """transformer.py:275-292 for one direction.  q, k: (..., N, D); vt: (..., N, DV) -> (..., DV, DV).

    attn = (q @ k^T) * scale; A = softmax(attn, -1) * softmax(attn, -2); F = (vt^T @ A) @ vt.
    """
    import torch
    td = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
    q, k, vt = (torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to(td) for a in (q, k, vt))
    attn = (q @ k.transpose(-1, -2)) * scale                      # torch-CPU ops: the reference's own operators,
    A = attn.softmax(dim=-1) * attn.softmax(dim=-2)               # multi-threaded (this is the cpu_baseline's hot spot)
    F = (vt.transpose(-1, -2) @ A) @ vt
    return F.numpy(), A.numpy()
