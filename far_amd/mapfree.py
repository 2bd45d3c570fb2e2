"""Map-free (mapfree_6dreg) counterparts of the pieces of FAR's hot path that live outside mp3d_loftr (SURVEY.md section
8 f4): the essential-matrix pose solver the regression model calls per sample, the per-sample matcher + solver loop of
RegressionModel.forward, and the correlation-volume warp of its aggregator (kernel K12, far_amd.ops.corr_volume_warp).

Reference:
  lib/models/matching/pose_solver.py:20-97     EssentialMatrixSolver.estimate_pose
  lib/models/regression/model.py:167-188       match_no_load (matcher on one pair -> mkpts0_f / mkpts1_f)
  lib/models/regression/model.py:236-273       the loop `for i in range(len(data['image0']))`: match, solve, pack
                                               loftr_rt (B, 3, 4) / inliers (B, 3 | 1), identity fallback
  lib/models/regression/aggregator.py:44-115   CorrelationVolumeWarping.forward

The matcher of the reference is upstream LoFTR (an un-vendored submodule, absent from /root/reference); its arithmetic is
the mp3d_loftr fork's, so far_amd.loftr.LoFTR with regress_rt = False stands in.  The minimal
solver is the GPU 8-point in every branch, batched over the pairs instead of looped.
"""
import numpy as np
import torch

from . import ops
from .solver import _branch, prior_point_cloud


class EssentialMatrixSolver:
    """pose_solver.py:20-97.  cfg: an object with EMAT_RANSAC.PIX_THRESHOLD / .CONFIDENCE (or None: the values of
    config/matching/mapfree/loftr_emat_*.yaml: 2.0 px, 0.9999)."""

    def __init__(self, cfg=None, use_prior_ransac=False, H=2048, seed=0, minimal=8):
        er = getattr(cfg, 'EMAT_RANSAC', None) if cfg is not None else None
        self.ransac_pix_threshold = float(er.PIX_THRESHOLD) if er is not None else 2.0
        self.ransac_confidence = float(er.CONFIDENCE) if er is not None else 0.9999
        self.use_prior_ransac = use_prior_ransac
        self.H, self.seed = H, seed
        self.minimal = minimal        # 8: normalized 8-point hypotheses; 5: Nister five-point (what Map-free executes through cv2, :81)
        self.mask = None

    def solve_batch(self, kpts0, kpts1, counts, K0, K1, priorRT=None):
        """kpts0 / kpts1: (Mtot, 2) fp32 GPU, pairs concatenated in order; counts: per-pair M (host); K0, K1 (B, 3, 3);
        priorRT: None or (B, 3, 4).  Returns far_amd.ops.solve_pose_batch's dict of device tensors."""
        dev = kpts0.device
        counts = [int(c) for c in counts]
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        K0d, K1d = K0.to(device=dev, dtype=torch.float64), K1.to(device=dev, dtype=torch.float64)
        mode = _branch('prior_ransac' if self.use_prior_ransac else 'ransac', priorRT is not None)
        if mode == 'prior':
            inl_th = torch.full((len(counts),), 3e-7, dtype=torch.float64, device=dev)                    # :63
            prior = torch.as_tensor(np.asarray(priorRT), dtype=torch.float32).reshape(-1, 3, 4).to(dev).contiguous()
            pcl = prior_point_cloud(dev)
        else:
            f = (K0d[:, 0, 0] + K1d[:, 1, 1] + K0d[:, 1, 1] + K1d[:, 0, 0]) / 4                          # :44
            inl_th = (self.ransac_pix_threshold / f) ** 2
            prior = pcl = None
        return ops.solve_pose_batch(kpts0.float().contiguous(), kpts1.float().contiguous(), offs, K0d, K1d,
                                    inl_th.contiguous(), mode == 'prior', priorRT=prior, pcl=pcl, prior_lambda=0.3,
                                    H=self.H, seed=self.seed, minimal=self.minimal)

    def estimate_pose(self, kpts0, kpts1, data, priorRT=None):
        """Single pair, the reference's contract: ((R (3,3), t (3,), n_inliers), inliers_best_tight, inliers_best_ultra_tight)
        as numpy / python numbers; identity + zeros when there are too few correspondences or no model (:31-34, :85-86)."""
        R, t = np.eye(3), np.zeros(3)
        if len(kpts0) < 5:
            return (R, t, 0), 0, 0
        dev = data['K_color0'].device if data['K_color0'].is_cuda else torch.device('cuda')
        k0 = torch.as_tensor(kpts0, dtype=torch.float32, device=dev)
        k1 = torch.as_tensor(kpts1, dtype=torch.float32, device=dev)
        out = self.solve_batch(k0, k1, [len(k0)], data['K_color0'].reshape(1, 3, 3), data['K_color1'].reshape(1, 3, 3),
                               None if priorRT is None else np.asarray(torch.as_tensor(priorRT).cpu())[None])
        self.mask = out['mask'].cpu().numpy()[:, None]
        if not int(out['status'][0]):
            return (R, t, 0), 0, 0
        return ((out['R'][0].cpu().numpy(), out['t'][0].cpu().numpy(), int(out['n_cheir'][0])),
                int(out['tight'][0]), int(out['ultra'][0]))


@torch.no_grad()
def match_and_solve(matcher, data, solver, priorRT=None, use_prior=True):
    """model.py:245-273 without the per-sample loop: the matcher runs on the whole batch, the solver on all pairs at once.
    data: image0 / image1 (B, 1, H, W) grayscale as the matcher takes them, K_color0 / K_color1 (B, 3, 3).
    Writes data['loftr_rt'] (B, 3, 4) float32 (identity where the solver found nothing, :268-269) and data['inliers']
    ((B, 3) [n, tight, ultra] with a prior-capable model, (B, 1) otherwise, :259-266), plus mkpts0_f / mkpts1_f / m_bids."""
    batch = {'image0': data['image0'], 'image1': data['image1']}
    matcher(batch)                                                                                   # :167-172
    B = data['image0'].shape[0]
    dev = data['image0'].device
    bids = batch['m_bids']
    order = torch.argsort(bids, stable=True)
    mk0, mk1 = batch['mkpts0_f'][order], batch['mkpts1_f'][order]
    counts = torch.bincount(bids, minlength=B).cpu()
    out = solver.solve_batch(mk0, mk1, counts.tolist(), data['K_color0'], data['K_color1'], priorRT)
    ok = out['status'].bool()
    eye = torch.eye(3, 4, dtype=torch.float32, device=dev).expand(B, 3, 4)
    rt = torch.cat([out['R'].float(), out['t'].float()[:, :, None]], dim=2)
    data['loftr_rt'] = torch.where(ok[:, None, None], rt, eye)
    n = torch.where(ok, out['n_cheir'], torch.zeros_like(out['n_cheir'])).float()
    if use_prior:
        z = torch.zeros_like(n)
        data['inliers'] = torch.stack([n, torch.where(ok, out['tight'].float(), z), torch.where(ok, out['ultra'].float(), z)], 1)
    else:
        data['inliers'] = n[:, None]
    data.update(mkpts0_f=mk0, mkpts1_f=mk1, m_bids=bids[order], match_counts=counts, solver_status=out['status'])
    return data


# ---------------------------------------------------------------------------------------------------------------------
# The matcher Map-free instantiates: upstream LoFTR (zju3dv/LoFTR, an un-vendored submodule: etc/feature_matching_baselines/LoFTR,
# .gitmodules) built from its `default_cfg` and loaded from a released checkpoint with strict=False
# (lib/models/regression/model.py:103-106).  The mp3d_loftr fork this build mirrors keeps upstream's module tree for the
# matcher (backbone / pos_encoding / loftr_coarse / coarse_matching / fine_preprocess / loftr_fine / fine_matching: same
# parameter names and shapes per layer), so an upstream state dict loads into far_amd.loftr.LoFTR once the config says what
# upstream's says: four (self, cross) coarse layer pairs, the pre-fix position encoding, no regression head.
# ---------------------------------------------------------------------------------------------------------------------
def upstream_loftr_config():
    """Upstream LoFTR's `default_cfg` (src/loftr/utils/cvpr_ds_config.py, lower-cased) as the dict far_amd.loftr.LoFTR takes."""
    from .config import far_eval_config
    cfg = far_eval_config()
    cfg['coarse'].update(layer_names=['self', 'cross'] * 4, temp_bug_fix=False)     # released checkpoints predate the fix
    cfg['match_coarse'].update(train_coarse_percent=0.4, skh_prefilter=True)
    cfg.update(regress_rt=False, solver='ransac', use_many_ransac_thr=False, regress_loftr_layers=0)
    return cfg


def load_upstream_loftr(state_dict, device='cuda'):
    """far_amd.loftr.LoFTR with upstream LoFTR's released weights: `state_dict` is the checkpoint's ['state_dict'] (keys
    'matcher.*', as torch.load(weights_path)['state_dict'] in model.py:105).  Loaded the way the reference does, strict=False:
    the optimal-transport checkpoints ('outdoor_ot.ckpt', the file Map-free names) carry `coarse_matching.bin_score`, which
    the dual-softmax matcher of `default_cfg` does not have.  Anything ELSE missing or unexpected is an error here -- a
    silent partial load would give a matcher that runs and matches nothing."""
    from .loftr import LoFTR
    m = LoFTR(upstream_loftr_config()).eval()
    sd = state_dict.get('state_dict', state_dict)
    res = m.load_state_dict(dict(sd), strict=False)
    unexpected = [k for k in res.unexpected_keys if not k.endswith('coarse_matching.bin_score')]
    if res.missing_keys or unexpected:
        raise KeyError(f'upstream LoFTR checkpoint does not fit: missing {res.missing_keys[:5]}{"..." if len(res.missing_keys) > 5 else ""}, '
                       f'unexpected {unexpected[:5]}{"..." if len(unexpected) > 5 else ""}')
    return m.to(device)
