"""Replay of the reference's evaluation call order on the far_amd modules
(mp3d_loftr/src/lightning/lightning_loftr.py:325-343, PL_LoFTR.test_step):

    matcher(batch) -> compute_supervision_RT(batch)
      -> [forward_rt_prediction(batch) -> compute_supervision_RT(batch)] x (FINE_PRED_STEPS - 1)
      -> forward_rt_prediction(batch)

This is the harness bench.py, smoke() and the end-to-end tests drive; the Lightning orchestration itself is
out of scope and keeps working against the same module interface.
"""
import torch

from .config import RunCfg
from .supervision import compute_supervision_RT


@torch.no_grad()
def test_step(matcher, batch, run_cfg=None, H=2048, seed=0):
    cfg = run_cfg or RunCfg(matcher.config['solver'], matcher.config.get('fine_pred_steps', 2))
    matcher(batch)                                                                  # :328
    batch['translation_scale'] = None                                               # :335
    compute_supervision_RT(batch, cfg, H=H, seed=seed)                              # :336
    steps = cfg.LOFTR.FINE_PRED_STEPS
    for i in range(steps):                                                          # :338
        matcher.forward_rt_prediction(batch)                                        # :340
        if i < steps - 1 and 'prior_ransac' in cfg.LOFTR.SOLVER:                    # :342
            compute_supervision_RT(batch, cfg, H=H, seed=seed)                      # :343
    return batch


@torch.no_grad()
def cached_step(matcher, batch, run_cfg=None, H=2048, seed=0):
    """BASELINE configs[3], the cached-prediction path (`--from_saved_preds`): the matcher is not run; the batch carries
    the cached transformer features (featmap0/1) and fine correspondences (mkpts0_f / mkpts1_f / m_bids, far_amd.cache_io).
    The reference reads the solver pose from disk and runs the head only (lightning_loftr.py:326, :334); here the GPU
    solver runs on the cached correspondences, so the same two solver rounds + two head calls as test_step execute:

        compute_supervision_RT -> [forward_rt_prediction -> compute_supervision_RT] x (FINE_PRED_STEPS - 1) -> forward_rt_prediction
    """
    cfg = run_cfg or RunCfg(matcher.config['solver'], matcher.config.get('fine_pred_steps', 2))
    batch['translation_scale'] = None
    batch.pop('priorRT', None)
    compute_supervision_RT(batch, cfg, H=H, seed=seed)
    steps = cfg.LOFTR.FINE_PRED_STEPS
    for i in range(steps):
        matcher.forward_rt_prediction(batch)
        if i < steps - 1 and 'prior_ransac' in cfg.LOFTR.SOLVER:
            compute_supervision_RT(batch, cfg, H=H, seed=seed)
    return batch
