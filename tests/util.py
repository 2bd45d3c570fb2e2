"""Shared synthetic-input builders for the parity tests (seeded, no reference access)."""
import numpy as np


def correlated_features(N, hw, C, seed=0, amp=1.2, noise=0.1, frac=1.0):
    """f1 = f0[perm] + noise: gives a dense set of confident mutual matches."""
    rng = np.random.default_rng(seed)
    L = hw[0] * hw[1]
    f0 = (amp * rng.standard_normal((N, L, C))).astype(np.float32)
    f1 = np.empty_like(f0)
    perms = []
    for n in range(N):
        perm = rng.permutation(L)
        perms.append(perm)
        f1[n] = f0[n][perm]
        if frac < 1.0:
            k = int(L * (1 - frac))
            bad = rng.choice(L, k, replace=False)
            f1[n][bad] = (amp * rng.standard_normal((k, C))).astype(np.float32)
    f1 = (f1 + noise * rng.standard_normal(f1.shape)).astype(np.float32)
    return f0, f1, perms


def two_view_scene(M, seed=0, outlier_frac=0.3, noise_px=0.3, K=None):
    """Synthetic calibrated two-view correspondences (pixels, float32) with ground-truth pose.
    Returns kpts0, kpts1 (M,2) float32, K (3,3) float64, R_gt, t_gt (unit)."""
    rng = np.random.default_rng(seed)
    if K is None:
        K = np.array([[517.97, 0, 320.], [0, 517.97, 240.], [0, 0, 1.]])
    ang = rng.uniform(-0.4, 0.4, 3)
    cx, sx = np.cos(ang[0]), np.sin(ang[0])
    cy, sy = np.cos(ang[1]), np.sin(ang[1])
    cz, sz = np.cos(ang[2]), np.sin(ang[2])
    R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
         @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    t = rng.uniform(-1, 1, 3)
    t[2] *= 0.3
    X = np.stack([rng.uniform(-3, 3, 4 * M), rng.uniform(-2, 2, 4 * M), rng.uniform(2, 8, 4 * M)], 1)
    X2 = X @ R.T + t
    p0 = X @ K.T
    p0 = p0[:, :2] / p0[:, 2:]
    p1 = X2 @ K.T
    p1 = p1[:, :2] / p1[:, 2:]
    ok = (X2[:, 2] > 0.5) & (p0[:, 0] > 0) & (p0[:, 0] < 640) & (p0[:, 1] > 0) & (p0[:, 1] < 480) \
        & (p1[:, 0] > 0) & (p1[:, 0] < 640) & (p1[:, 1] > 0) & (p1[:, 1] < 480)
    p0, p1 = p0[ok][:M], p1[ok][:M]
    M = len(p0)
    p1 = p1 + noise_px * rng.standard_normal(p1.shape)
    nout = int(outlier_frac * M)
    if nout:
        idx = rng.choice(M, nout, replace=False)
        p1[idx] = np.stack([rng.uniform(0, 640, nout), rng.uniform(0, 480, nout)], 1)
    return p0.astype(np.float32), p1.astype(np.float32), K, R, t / np.linalg.norm(t)



def deviation(name, got, ref, atol, rtol=0.0):
    """assert_allclose that also PRINTS the measured deviation (max |got - ref|, and relative to max |ref|), so that
    the bars in the tests can be kept at ~3x what the hardware run measures (`pytest -s` shows the lines)."""
    import torch
    g = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    r = ref.detach().cpu().numpy() if torch.is_tensor(ref) else np.asarray(ref)
    d = float(np.abs(g.astype(np.float64) - r.astype(np.float64)).max()) if g.size else 0.0
    scale = float(np.abs(r).max()) if r.size else 1.0
    print(f'[deviation] {name}: max|d| = {d:.3e}  (max|ref| = {scale:.3e}, rel = {d / max(scale, 1e-30):.3e}; bar atol {atol:g} rtol {rtol:g})')
    import os
    if os.environ.get('FAR_MEASURE_ONLY') != '1':          # measurement runs print every deviation without stopping
        np.testing.assert_allclose(g, r, atol=atol, rtol=rtol, err_msg=name)
    return d
