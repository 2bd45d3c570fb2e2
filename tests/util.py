"""Shared synthetic-input builders for the parity tests (seeded, no reference access).  tools/make_goldens.py imports the
same builders when it runs the reference, so a golden's inputs and the test's inputs are one piece of code."""
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


# ---------------------------------------------------------------------------------------------------------------------
# inputs shared with tools/make_goldens.py (G10, G14, G15)
# ---------------------------------------------------------------------------------------------------------------------
def train_inputs():
    """Shared by the generator and tests/test_training_cpu.py: one synthetic pair, GT coarse matches from the known
    band disparities, fake solver outputs."""
    from far_amd import synth
    im0, im1 = synth.synth_image_pair(1, seed=0)
    disp = (8, 40, 72)
    ii, jj = [], []
    for y in range(60):
        d = disp[min(2, (y * 8) // 160)] // 8
        for x in range(80):
            if 0 <= x - d < 80:
                ii.append(y * 80 + x)
                jj.append(y * 80 + x - d)
    ang = 0.3
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    rt = np.concatenate([R, np.array([0.6, -0.1, 0.79])[:, None]], 1)
    return im0, im1, np.array(ii, np.int64), np.array(jj, np.int64), rt


GRAD_KEYS = ['backbone.conv1.weight', 'backbone.layer1.0.conv1.weight', 'backbone.layer3.1.bn2.weight',
             'backbone.layer3_outconv.weight', 'backbone.layer1_outconv2.3.weight',
             'loftr_coarse.layers.0.q_proj.weight', 'loftr_coarse.layers.5.mlp.2.weight', 'loftr_coarse.layers.3.norm1.bias',
             'fine_preprocess.merge_feat.weight', 'loftr_fine.layers.1.v_proj.weight',
             'loftr_regress.loftr.layers.0.k_proj.weight', 'loftr_regress.emm.cross_attn.qkv.weight',
             'loftr_regress.emm.pos_embed', 'loftr_regress.encoder.0.weight', 'loftr_regress.moe_predictor.4.weight',
             'loftr_regress.pose_regressor_simple_moe.2.bias']


def train_step(m, im0, im1, ii, jj, rt):
    """One training-mode forward + backward with a synthetic loss touching conf_matrix, expec_f and regressed_rt."""
    import torch
    data = {'image0': torch.from_numpy(im0), 'image1': torch.from_numpy(im1),
            'spv_b_ids': torch.zeros(len(ii), dtype=torch.int64), 'spv_i_ids': torch.from_numpy(ii),
            'spv_j_ids': torch.from_numpy(jj)}
    m.train()
    torch.manual_seed(123)
    m(data, train=True)
    data.update({'loftr_rt': torch.from_numpy(rt), 'num_correspondences': torch.tensor([731]),
                 'num_correspondences_before_ransac': torch.tensor([1500]), 'inliers_best_tight': torch.tensor([410]),
                 'inliers_best_ultra_tight': torch.tensor([57])})
    m.forward_rt_prediction(data)
    conf = data['conf_matrix']
    loss_c = -torch.log(conf[0, data['spv_i_ids'], data['spv_j_ids']] + 1e-6).mean()
    loss_f = data['expec_f'].pow(2).mean()
    loss_rt = data['regressed_rt'].pow(2).sum()
    loss = loss_c + loss_f + loss_rt
    m.zero_grad()
    loss.backward()
    return data, (loss_c, loss_f, loss_rt)


def spvs_scene(seed=61, N=2, H=480, W=640):
    """Shared by the generator and tests: two fronto-parallel-ish depth maps per pair related by a known rigid motion
    (planes at different depths per image band, a few zero-depth holes), Matterport intrinsics."""
    rng = np.random.default_rng(seed)
    K = np.array([[517.97, 0, 320.], [0, 517.97, 240.], [0, 0, 1.]], np.float32)
    depth0 = np.empty((N, H, W), np.float32)
    depth1 = np.empty((N, H, W), np.float32)
    T01 = np.zeros((N, 4, 4), np.float32)
    for n in range(N):
        ang = 0.05 * (n + 1)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
        t = np.array([0.15 * (n + 1), 0.02, 0.05], np.float32)
        T01[n, :3, :3] = R; T01[n, :3, 3] = t; T01[n, 3, 3] = 1
        # a slanted plane n.X = d in camera 0; render its depth in both cameras analytically
        nrm = np.array([0.1, -0.05, 1.0], np.float32); nrm /= np.linalg.norm(nrm)
        d = 3.0 + 0.5 * n
        ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
        rays = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], -1)
        depth0[n] = d / (rays @ nrm)
        # plane in camera 1: n1 = R n, d1 = d + n1 . t
        n1 = R @ nrm; d1 = d + n1 @ t
        depth1[n] = d1 / (rays @ n1)
        holes = rng.integers(0, H // 8, (20, 2))
        for hy, hx in holes:
            depth0[n, hy * 8:hy * 8 + 8, (hx * 8) % W:(hx * 8) % W + 8] = 0
    T10 = np.linalg.inv(T01).astype(np.float32)
    return depth0, depth1, T01, T10, np.stack([K] * N)


def loss_inputs(seed=71, N=2, L=48, M=40):
    """Shared by the generator and the tests: random coarse confidences / ground truth / fine predictions / poses."""
    rng = np.random.default_rng(seed)
    conf = rng.uniform(0, 1, (N, L, L)).astype(np.float32) ** 3
    conf[0, 3, 5] = 0.0                                   # clamp paths (:84)
    conf[1, 7, 7] = 1.0
    gt = np.zeros((N, L, L), np.float32)
    pos = [(0, 3, 5), (1, 7, 7)] + [(int(rng.integers(N)), int(rng.integers(L)), int(rng.integers(L))) for _ in range(30)]
    for b, i, j in pos:
        gt[b, i, j] = 1
    expec_f = np.concatenate([rng.normal(0, 0.5, (M, 2)), rng.uniform(0.05, 2.0, (M, 1))], 1).astype(np.float32)
    expec_f_gt = rng.normal(0, 0.7, (M, 2)).astype(np.float32)
    expec_rt = rng.normal(0, 1, 9).astype(np.float32)
    ang = 0.4
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    T[:3, 3] = [0.3, -0.2, 1.1]
    T = np.stack([T, np.eye(4, dtype=np.float32)])
    # spvs_fine inputs
    w_pt0 = rng.uniform(0, 640, (N, L, 2)).astype(np.float32)
    pt1 = rng.uniform(0, 640, (N, L, 2)).astype(np.float32)
    b_ids = rng.integers(0, N, M)
    i_ids = rng.integers(0, L, M)
    j_ids = rng.integers(0, L, M)
    return dict(conf=conf, gt=gt, expec_f=expec_f, expec_f_gt=expec_f_gt, expec_rt=expec_rt, T=T, w_pt0=w_pt0, pt1=pt1,
                b_ids=b_ids, i_ids=i_ids, j_ids=j_ids)


def eval_batch(seed=81, B=4, M=60):
    """Inputs of golden G16 (shared with tools/make_goldens.py): B pairs of synthetic two-view correspondences with ground
    truth, a head output per pair, committed solver fits (pair 1 fails) and a prior."""
    rng = np.random.default_rng(seed)
    T, mk0, mk1, bids, fit_R, fit_t = [], [], [], [], [], []
    K = None
    for b in range(B):
        p0, p1, K, R, t = two_view_scene(M + 7 * b, seed=seed + b, outlier_frac=0.25)
        Tb = np.eye(4, dtype=np.float32)
        Tb[:3, :3] = R
        Tb[:3, 3] = t * rng.uniform(0.5, 2.5)
        T.append(Tb)
        mk0.append(p0)
        mk1.append(p1)
        bids.append(np.full(len(p0), b, np.int64))
        _, _, _, Re, te = two_view_scene(8, seed=seed + b + 100 * (b % 2))          # a fit near (even b) / off (odd b) the truth
        fit_R.append(Re)
        fit_t.append(te * (1.0 if b % 3 else -1.0))
    m_bids = np.concatenate(bids)
    K32 = np.stack([K.astype(np.float32)] * B)
    regressed = rng.normal(0, 0.6, (B, 9)).astype(np.float32)
    ang = 0.2
    prior = np.concatenate([np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]),
                            np.array([[0.5], [0.1], [-0.8]])], 1)
    return dict(T=np.stack(T), K0=K32, K1=K32.copy(), m_bids=m_bids, mk0=np.concatenate(mk0), mk1=np.concatenate(mk1),
                regressed_rt=regressed, fit_R=np.stack(fit_R), fit_t=np.stack(fit_t), fit_ok=np.array([1, 0, 1, 1]),
                fit_mask=(rng.uniform(size=len(m_bids)) < 0.6).astype(np.uint8), priorRT=prior)


def eval_metrics_table(seed=82, n=23):
    """A gathered per-pair metrics table as PL_LoFTR.validation_epoch_end hands to aggregate_metrics: two duplicated
    identifiers (DistributedSampler padding), one pair without matches, failed fits."""
    rng = np.random.default_rng(seed)
    ids = [f'scene{i // 5}#im{i}a#im{i}b' for i in range(n)]
    ids[n - 2], ids[n - 1] = ids[0], ids[3]
    R = rng.gamma(1.2, 6.0, n)
    t = rng.gamma(1.5, 9.0, n)
    R[n - 2] = 0.7          # the duplicate's later entry is the one the de-duplication keeps
    epi = [rng.gamma(0.6, 6e-4, int(rng.integers(0, 40))) for _ in range(n)]
    epi[5] = np.zeros(0)
    return {'identifiers': ids, 'epi_errs': epi, 'R_errs': list(R), 't_errs': list(t), 't_errs_abs': list(rng.gamma(2.0, 0.5, n)),
            'successful_fits': list((rng.uniform(size=n) < 0.8).astype(np.int64))}


def planar_scene(M, seed, kind, noise=0.3, outl=0.2, K=None):
    """Two-view correspondences (pixels, float32) of a scene that is 'general' (depths 2..8), 'two_planes' (a wall and a
    floor) or one slanted 'plane' -- the configuration on which the normalized 8-point is degenerate.  Returns kpts0, kpts1,
    R_gt, t_gt (unit)."""
    rng = np.random.default_rng(seed)
    if K is None:
        K = np.array([[517.97, 0, 320.], [0, 517.97, 240.], [0, 0, 1.]])
    ang = rng.uniform(-0.3, 0.3, 3)
    cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
         @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    t = rng.uniform(-1, 1, 3)
    t[2] *= 0.3
    n = 6 * M + 40
    xy = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n)], 1)
    if kind == 'general':
        z = rng.uniform(2, 8, n)
    elif kind == 'plane':
        nrm = np.array([0.2, -0.1, 1.0])
        z = (4.0 - nrm[0] * xy[:, 0] - nrm[1] * xy[:, 1]) / nrm[2]
    elif kind == 'two_planes':
        wall = rng.uniform(size=n) < 0.7
        z = np.where(wall, 5.0 + 0.3 * xy[:, 0], rng.uniform(2, 6, n))
        xy[~wall, 1] = 1.5
    else:
        raise ValueError(kind)
    X = np.stack([xy[:, 0], xy[:, 1], z], 1)
    X2 = X @ R.T + t
    p0 = X @ K.T
    p0 = p0[:, :2] / p0[:, 2:]
    p1 = X2 @ K.T
    p1 = p1[:, :2] / p1[:, 2:]
    ok = (X2[:, 2] > 0.5) & (X[:, 2] > 0.5) & (p0[:, 0] > 0) & (p0[:, 0] < 640) & (p0[:, 1] > 0) & (p0[:, 1] < 480) \
        & (p1[:, 0] > 0) & (p1[:, 0] < 640) & (p1[:, 1] > 0) & (p1[:, 1] < 480)
    p0, p1 = p0[ok][:M], p1[ok][:M]
    M = len(p0)
    p1 = p1 + noise * rng.standard_normal(p1.shape)
    no = int(outl * M)
    if no:
        idx = rng.choice(M, no, replace=False)
        p1[idx] = np.stack([rng.uniform(0, 640, no), rng.uniform(0, 480, no)], 1)
    return p0.astype(np.float32), p1.astype(np.float32), R, t / np.linalg.norm(t)


def masked_coarse_inputs(seed=91, N=2, h=24, w=32, C=256):
    """Shared by tools/make_goldens.py:g18 and the tests: coarse feature maps of a padded-mask batch (images of different sizes
    padded to one h x w grid, as MegaDepth / ScanNet-style batches are): correlated features, validity masks, sparse ground-truth
    ids inside the valid regions."""
    rng = np.random.default_rng(seed)
    L = h * w
    f0 = (1.2 * rng.standard_normal((N, L, C))).astype(np.float32)
    f1 = np.zeros_like(f0)
    ii, jj, bb = [], [], []
    valid0 = [(h, w - 8), (h - 6, w)]
    valid1 = [(h - 4, w), (h, w - 10)]
    m0 = np.zeros((N, h, w), bool)
    m1 = np.zeros((N, h, w), bool)
    for n in range(N):
        m0[n, :valid0[n][0], :valid0[n][1]] = True
        m1[n, :valid1[n][0], :valid1[n][1]] = True
        perm = rng.permutation(L)
        f1[n] = f0[n][perm] + 0.15 * rng.standard_normal((L, C)).astype(np.float32)
        inv = np.argsort(perm)                               # f1[j] ~ f0[perm[j]]  ->  i = perm[j]
        for j in range(L):
            i = perm[j]
            if m0[n].reshape(-1)[i] and m1[n].reshape(-1)[j] and rng.uniform() < 0.5:
                bb.append(n); ii.append(i); jj.append(j)
    order = np.lexsort((np.array(ii), np.array(bb)))
    return {'f0': f0, 'f1': f1, 'mask0': m0, 'mask1': m1, 'spv_b_ids': np.array(bb, np.int64)[order],
            'spv_i_ids': np.array(ii, np.int64)[order], 'spv_j_ids': np.array(jj, np.int64)[order], 'h': h, 'w': w}


# ---- golden G19 (the 8-Point-ViT shape of K2): the seeded weights / input both tools/make_golden_vit.py and the tests replay
VIT_INTRINSICS = (13.7, 14.2, 12.0, 12.0)          # fx, fy, cx, cy on the 24 x 24 feature grid (model.py:120-129 rescales to it)


def vit_seeded_fill(block, seed):
    """Fills a CrossBlock-shaped module (parameters visited in sorted-name order) and returns the (2, 576, 192) input of one pair,
    all from one seeded CPU generator: identical on every machine (torch's CPU generator is)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(block.named_parameters()):
            if name.endswith('norm1.weight') or name.endswith('norm2.weight'):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (1.5 / p.shape[1] ** 0.5))
        f0 = torch.randn(1, 576, 192, generator=g)
        f1 = 0.5 * f0 + torch.randn(1, 576, 192, generator=g)
    return torch.cat([f0, f1], 0)
