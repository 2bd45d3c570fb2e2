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
from .supervision import compute_supervision_RT, compute_supervision_coarse, compute_supervision_fine


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


def train_step(matcher, batch, loss_fn, run_cfg=None, H=2048, seed=0, forward=None):
    """BASELINE configs[2]: one training forward in the reference's order (PL_LoFTR._trainval_inference,
    lightning_loftr.py:129-172, with training_step :174-182 left to the caller: batch['loss'].backward(), optimizer):

        compute_supervision_coarse -> matcher(batch, train=True) -> compute_supervision_fine -> compute_supervision_RT
          -> no_grad[forward_rt_prediction -> compute_supervision_RT] x (FINE_PRED_STEPS - 1) -> forward_rt_prediction -> loss

    On GPU tensors the matcher's training forward runs K1's sparse-position kernels, K5, K9 (Linear) and K2 with their
    HIP backward kernels; the ground truth never becomes a dense conf_matrix_gt (far_amd/losses.py).  A batch that already
    carries spv_b_ids / spv_i_ids / spv_j_ids (no depth maps) skips the coarse supervision, as the reference does for
    its depth-less data source (:131-133).  `forward`: the DistributedDataParallel wrapper of `matcher` when there is one
    (its forward arms the gradient all-reduce hooks; the head call below runs on the wrapped module as in the reference)."""
    cfg = run_cfg or RunCfg(matcher.config['solver'], matcher.config.get('fine_pred_steps', 2))
    if 'depth0' in batch:
        compute_supervision_coarse(batch, cfg)                                      # :133
    (forward or matcher)(batch, train=True)                                         # :136
    if 'spv_w_pt0_i' in batch:
        compute_supervision_fine(batch, cfg)                                        # :140
    batch.update(num_correspondences_before_ransac=0, num_correspondences_after_ransac=0)   # :142-145
    if matcher.config['regress_rt']:
        batch['translation_scale'] = None                                           # :156
        with torch.no_grad():
            compute_supervision_RT(batch, cfg, H=H, seed=seed)                      # :157
        steps = cfg.LOFTR.FINE_PRED_STEPS
        for i in range(steps):                                                      # :159
            if i < steps - 1 and 'prior_ransac' in cfg.LOFTR.SOLVER:
                with torch.no_grad():                                               # :161-164
                    matcher.forward_rt_prediction(batch)
                    compute_supervision_RT(batch, cfg, H=H, seed=seed)
            else:
                matcher.forward_rt_prediction(batch)                                # :167
                loss_fn(batch)                                                      # :169
    else:
        loss_fn(batch)                                                              # :172
    return batch
