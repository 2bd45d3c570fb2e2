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
