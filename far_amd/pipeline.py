"""Replay of the reference's step functions on the far_amd modules: test_step (PL_LoFTR.test_step,
mp3d_loftr/src/lightning/lightning_loftr.py:325-343), cached_step (the --from_saved_preds form of it), train_step
(training_step -> _trainval_inference :129-172), val_step (validation_step :266-281 = _trainval_inference with
train=False + _compute_metrics :227-264).  The evaluation order:

    matcher(batch) -> compute_supervision_RT(batch)
      -> [forward_rt_prediction(batch) -> compute_supervision_RT(batch)] x (FINE_PRED_STEPS - 1)
      -> forward_rt_prediction(batch)

This is the harness bench.py, smoke() and the end-to-end tests drive; the Lightning orchestration itself is
out of scope and keeps working against the same module interface.
"""
import torch

from .config import RunCfg
from .supervision import compute_supervision_RT, compute_supervision_coarse, compute_supervision_fine


_MISSING = object()


def _one_guard(matcher, batch, seq, device_key):
    """Runs seq() -- a re-runnable sequence of matcher / solver / head calls on `batch` -- under ONE activation-range guard of the
    LoFTR module (LoFTR.guarded_sequence: one host read of the overflow flag behind the whole step instead of one per call, so that
    the host can enqueue the whole step behind the match-count read without waiting for the GPU three more times).  A caller-supplied
    priorRT is restored before a re-run; wrappers (DistributedDataParallel) are looked through; anything without the guard runs plain."""
    core = getattr(matcher, 'module', matcher)
    t = batch.get(device_key)
    if not hasattr(core, 'guarded_sequence') or not torch.is_tensor(t) or not t.is_cuda:
        return seq()
    prior0, pdev0 = batch.get('priorRT', _MISSING), batch.get('_priorRT_device', _MISSING)

    again = [False]                                         # (not `run.again`: a function that refers to itself is a reference cycle that
                                                            # keeps `batch` -- the step's device tensors -- alive until the cyclic collector runs)

    def run():
        if again[0]:                                        # a re-run at a wider range starts from what the caller handed in
            for k, v in (('priorRT', prior0), ('_priorRT_device', pdev0)):
                if v is _MISSING:
                    batch.pop(k, None)
                else:
                    batch[k] = v
            core.invalidate_head_cache(batch)
        again[0] = True
        return seq()
    return core.guarded_sequence(run, t.device, tuple(batch[k] for k in ('image0', 'image1') if k in batch))


@torch.no_grad()
def test_step(matcher, batch, run_cfg=None, H=2048, seed=0):
    cfg = run_cfg or RunCfg(matcher.config['solver'], matcher.config.get('fine_pred_steps', 2))

    def seq():
        if cfg.LOFTR.FINE_PRED_STEPS > 0:
            # the head WILL be called on this batch: its feature stage may be enqueued behind the coarse matcher (LoFTR.head_prefetch)
            batch['_far_head_follows'] = True
        try:
            out = matcher(batch)                                                    # :328
        finally:
            batch.pop('_far_head_follows', None)           # also when the matcher raised: a later matcher-only call must not prefetch
        if isinstance(out, dict) and out is not batch:     # a wrapper that copied the dict (DDP with device_ids): merge its writes
            batch.update(out)
            batch.pop('_far_head_follows', None)
        batch['translation_scale'] = None                                           # :335
        compute_supervision_RT(batch, cfg, H=H, seed=seed)                          # :336
        steps = cfg.LOFTR.FINE_PRED_STEPS
        for i in range(steps):                                                      # :338
            matcher.forward_rt_prediction(batch)                                    # :340
            if i < steps - 1 and 'prior_ransac' in cfg.LOFTR.SOLVER:                # :342
                compute_supervision_RT(batch, cfg, H=H, seed=seed)                  # :343
        return batch
    return _one_guard(matcher, batch, seq, 'image0')


@torch.no_grad()
def cached_step(matcher, batch, run_cfg=None, H=2048, seed=0):
    """BASELINE configs[3], the cached-prediction path (`--from_saved_preds`): the matcher is not run; the batch carries
    the cached transformer features (featmap0/1) and fine correspondences (mkpts0_f / mkpts1_f / m_bids, far_amd.cache_io).
    The reference reads the solver pose from disk and runs the head only (lightning_loftr.py:326, :334); here the GPU
    solver runs on the cached correspondences, so the same two solver rounds + two head calls as test_step execute:

        compute_supervision_RT -> [forward_rt_prediction -> compute_supervision_RT] x (FINE_PRED_STEPS - 1) -> forward_rt_prediction
    """
    cfg = run_cfg or RunCfg(matcher.config['solver'], matcher.config.get('fine_pred_steps', 2))
    batch.pop('priorRT', None)
    batch.pop('_priorRT_device', None)

    def seq():
        batch['translation_scale'] = None
        compute_supervision_RT(batch, cfg, H=H, seed=seed)
        steps = cfg.LOFTR.FINE_PRED_STEPS
        for i in range(steps):
            matcher.forward_rt_prediction(batch)
            if i < steps - 1 and 'prior_ransac' in cfg.LOFTR.SOLVER:
                compute_supervision_RT(batch, cfg, H=H, seed=seed)
        return batch
    return _one_guard(matcher, batch, seq, 'featmap0')


def _trainval_inference(matcher, batch, loss_fn, cfg, train, H, seed, forward=None):
    """PL_LoFTR._trainval_inference (lightning_loftr.py:129-172) for LOFTR.FROM_SAVED_PREDS = None and no correspondence
    transformer (every FAR script).  The reference keys the supervision calls on the data source (:131-140); so does this:
    'interiornet_streetlearn' has no depth and skips them, every other source gets compute_supervision_coarse from
    depth0 / depth1 -- or, for a depth-less caller that labels its own batches (synthetic data: far_amd.synth), the
    complete label set spv_b_ids / spv_i_ids / spv_j_ids / spv_w_pt0_i / spv_pt1_i already in the batch.  Anything in
    between is an error here, not a KeyError three calls later."""
    src = batch['dataset_name'][0].lower() if 'dataset_name' in batch else 'mp3d'
    supervised = src != 'interiornet_streetlearn'
    if supervised:                                                                  # :131-133
        if 'depth0' in batch:
            compute_supervision_coarse(batch, cfg)
        elif not all(k in batch for k in ('spv_b_ids', 'spv_i_ids', 'spv_j_ids', 'spv_w_pt0_i', 'spv_pt1_i')):
            raise KeyError(f"data source '{src}' is depth-supervised: the batch needs depth0 / depth1 / T_0to1 / T_1to0 "
                           f"(compute_supervision_coarse) or a complete precomputed label set spv_b_ids, spv_i_ids, "
                           f"spv_j_ids, spv_w_pt0_i, spv_pt1_i")
    out = (forward or matcher)(batch, train=train)                                  # :136
    if isinstance(out, dict) and out is not batch:
        # a wrapper that rebuilds dict arguments (DistributedDataParallel with device_ids does, through _recursive_to)
        # hands the module a COPY: the results the module wrote live in the returned dict
        batch.update(out)
    assert 'b_ids' in batch and 'mkpts0_f' in batch, 'the matcher wrote its results into a different dict than the caller\'s'
    if supervised:
        compute_supervision_fine(batch, cfg)                                        # :138-140
    batch.update(num_correspondences_before_ransac=0, num_correspondences_after_ransac=0)   # :142-145
    if matcher.config['regress_rt']:
        if matcher.config['regress']['use_simple_moe']:                             # :153-157
            batch['translation_scale'] = None
            with torch.no_grad():
                compute_supervision_RT(batch, cfg, H=H, seed=seed)
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


def train_step(matcher, batch, loss_fn, run_cfg=None, H=2048, seed=0, forward=None):
    """BASELINE configs[2]: one training forward in the reference's order (PL_LoFTR.training_step :229-230 ->
    _trainval_inference(batch, train=True), with the rest of training_step left to the caller: batch['loss'].backward(),
    optimizer):

        compute_supervision_coarse -> matcher(batch, train=True) -> compute_supervision_fine -> compute_supervision_RT
          -> no_grad[forward_rt_prediction -> compute_supervision_RT] x (FINE_PRED_STEPS - 1) -> forward_rt_prediction -> loss

    On GPU tensors the matcher's training forward runs K1's sparse-position kernels, K5, K9 (Linear) and K2 with their
    HIP backward kernels; the ground truth never becomes a dense conf_matrix_gt (far_amd/losses.py).  `forward`: the
    DistributedDataParallel wrapper of `matcher` when there is one (its forward arms the gradient all-reduce hooks; the
    head call runs on the wrapped module as in the reference)."""
    cfg = run_cfg or RunCfg(matcher.config['solver'], matcher.config.get('fine_pred_steps', 2))
    return _trainval_inference(matcher, batch, loss_fn, cfg, True, H, seed, forward)


def compute_metrics(batch, run_cfg, H=2048, seed=0):
    """PL_LoFTR._compute_metrics (lightning_loftr.py:227-264) for the matcher configurations: epipolar errors of every
    match, pose errors of every pair, and the per-pair table validation_epoch_end gathers (same keys; `identifiers`
    from batch['pair_names'] when the loader provides them)."""
    from .metrics import compute_pose_errors, compute_symmetrical_epipolar_errors
    compute_symmetrical_epipolar_errors(batch)                                      # :233
    compute_pose_errors(batch, run_cfg, H=H, seed=seed)                             # :239
    bs = batch['image0'].size(0) if 'image0' in batch else batch['K0'].shape[0]
    names = list(zip(*batch['pair_names'])) if 'pair_names' in batch else [(f'pair{b}',) for b in range(bs)]
    epi, mb = batch['epi_errs'], batch['m_bids']
    if mb.numel() > 1 and bool((mb[1:] < mb[:-1]).any()):
        mb = torch.sort(mb, stable=True)[0]                                         # epi_errs is in pair-after-pair order
    epi_h, mb_h = epi.cpu().numpy(), mb.cpu().numpy()
    metrics = {
        'identifiers': ['#'.join(names[b]) for b in range(bs)],                     # :248
        'epi_errs': [epi_h[mb_h == b] for b in range(bs)],                          # :249
        'R_errs': batch['R_errs'], 't_errs': batch['t_errs'], 't_errs_abs': batch['t_errs_abs'],
        'inliers': batch['inliers'], 'successful_fits': batch['successful_fits'],
        'gt_R': batch['T_0to1'][:, :3, :3].cpu(),
        'pred_R': torch.from_numpy(batch['pred_R']).unsqueeze(0).cpu(),
        'pred_t': torch.from_numpy(batch['pred_t']).unsqueeze(0).cpu(),
    }
    if 'lightweight_numcorr' in batch:
        metrics['lightweight_numcorr'] = [x.cpu().numpy() for x in batch['lightweight_numcorr']]
    return {'metrics': metrics}, names


@torch.no_grad()
def val_step(matcher, batch, loss_fn, run_cfg=None, H=2048, seed=0):
    """PL_LoFTR.validation_step (lightning_loftr.py:266-281, without the figures): _trainval_inference(batch) with
    train=False -- the supervision, the matcher in eval mode, both solver rounds, the head, THE LOSS -- then
    _compute_metrics.  In eval mode the coarse matcher does not build the (N, 4800, 4800) conf_matrix; it evaluates it at
    the ground-truth positions (data['conf_pos'], forward-only launch of K1's sparse-position kernel), which is all the
    loss reads.  Returns {'metrics': ..., 'loss_scalars': ...}."""
    cfg = run_cfg or RunCfg(matcher.config['solver'], matcher.config.get('fine_pred_steps', 2))
    _trainval_inference(matcher, batch, loss_fn, cfg, False, H, seed)
    ret, _ = compute_metrics(batch, cfg, H=H, seed=seed)
    return {**ret, 'loss_scalars': batch['loss_scalars']}
