"""The N > 1 path on real GPU work, as far as a one-GPU box allows: `bench.py --gpus 2` starts two ranks that BOTH use
cuda:0 (`--share-gpu`, gloo process group -- RCCL refuses two ranks on one device).  Functional only: launcher, rank
environment, sharded seeds, barrier / max-over-ranks / per-rank gather on device tensors, and for the training workload
DistributedDataParallel's gradient all-reduce through the HIP autograd Functions.  Timings of these runs mean nothing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra, gpus=2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--backend', 'gloo', '--share-gpu', '--steps', '1',
           '--warmup', '1', '--no-cpu-baseline'] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints the one JSON line
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_two_ranks_eval_step():
    res = _run(['--pairs', '2', '--hyp', '256'])
    assert res['n_gpus'] == 2 and len(res['per_rank_ms_per_step']) == 2
    assert res['value'] > 0 and res['config']['pairs_per_gpu'] == 2
    assert res['config']['solver_success_frac'] > 0.5
    assert 'cpu_baseline' not in res                               # N > 1 lines carry no CPU leg


@pytest.mark.timeout(1200)
def test_two_ranks_training_step_under_ddp():
    res = _run(['--workload', 'c3', '--pairs', '1', '--hyp', '256'])
    assert res['n_gpus'] == 2 and len(res['per_rank_ms_per_step']) == 2
    assert res['config']['parallelism'].startswith('ddp2')
    assert all(v == v and abs(v) < 1e4 for v in res['config']['losses'].values())      # finite losses after DDP steps


@pytest.mark.timeout(1800)
def test_eight_ranks_eval_step_rehearsal():
    """N = 8 rehearsed on the one GPU a box has: `bench.py --gpus 8 --share-gpu --backend gloo --pairs 1` -- eight
    launcher children (started before anything touches the GPU), eight seeds, the barrier / max-over-ranks / eight-way gather of the
    timing rule, one JSON line with n_gpus 8 and eight per-rank entries (step time and host CPU time).  Timings mean nothing here."""
    res = _run(['--pairs', '1', '--hyp', '256', '--no-other-modes', '--no-other-workloads', '--skip-rooflines'], gpus=8)
    assert res['n_gpus'] == 8 and len(res['per_rank_ms_per_step']) == 8 and len(res['per_rank_host_ms_per_step']) == 8
    assert res['value'] > 0 and res['config']['pairs_per_gpu'] == 1 and res['config']['parallelism'].startswith('dp8')
    assert res['host_ms_per_step'] == max(res['per_rank_host_ms_per_step']) > 0
    assert res['prime_steps'] <= res['prime_cap']
    assert res['process_group'] == {'backend': 'gloo', 'world_size': 8}


@pytest.mark.timeout(1800)
def test_eight_ranks_training_step_under_ddp_rehearsal():
    """The training workload at eight ranks on one GPU: DDP's bucketed gradient all-reduce across eight ranks through the HIP autograd
    Functions, priming steps whose stop decision is collective (bench.settle: every rank leaves after the same step count -- ranks
    that left at different counts would un-pair DDP's all-reduces and hang here)."""
    res = _run(['--workload', 'c3', '--pairs', '1', '--hyp', '256'], gpus=8)
    assert res['n_gpus'] == 8 and len(res['per_rank_ms_per_step']) == 8 and len(res['per_rank_host_ms_per_step']) == 8
    assert res['config']['parallelism'].startswith('ddp8')
    assert all(v == v and abs(v) < 1e4 for v in res['config']['losses'].values())
    assert res['prime_steps'] <= res['prime_cap']


@pytest.mark.timeout(900)
def test_ddp_built_with_device_ids_still_hands_the_results_back():
    """DistributedDataParallel(model, device_ids=[0]) rebuilds dict arguments on the way in (_recursive_to), so the module
    writes b_ids / conf_pos / expec_f / ... into a COPY of the caller's batch.  LoFTR.forward returns its dict and
    pipeline._trainval_inference merges a returned copy back: the reference-order training step must work for both DDP
    forms (with device_ids: the copy path; without: the pass-through path bench.py uses).  One process, world size 1."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from far_amd import parallel, synth
    from far_amd.config import RunCfg, far_train_config
    from far_amd.loftr import LoFTR
    from far_amd.losses import LoFTRLoss
    from far_amd.pipeline import train_step
    cfg = far_train_config()
    model = LoFTR(cfg['loftr'])
    synth.load_synthetic(model, seed=0)
    model = model.cuda().train()
    loss_fn = LoFTRLoss(cfg).train()
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{parallel.free_port()}', rank=0, world_size=1)
    try:
        base = synth.synth_training_batch(1, seed=9, device='cuda')
        losses = []
        for ids in ([0], None):
            fwd = DDP(model, device_ids=ids)
            batch = dict(base)
            torch.manual_seed(3)
            train_step(model, batch, loss_fn, RunCfg('prior_ransac', 2), H=256, seed=0, forward=fwd)
            assert 'b_ids' in batch and batch['conf_pos'].requires_grad and batch['loss'].requires_grad
            model.zero_grad()
            batch['loss'].backward()
            g = model.loftr_coarse.layers[0].q_proj.weight.grad
            assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
            losses.append(float(batch['loss'].detach()))
            del fwd
        np.testing.assert_allclose(losses[0], losses[1], rtol=1e-6)
    finally:
        dist.destroy_process_group()


def _run_rccl_world1(extra, launcher):
    """One rank that still joins an RCCL process group (`--dist-at-world-1 --backend nccl`): the child is started before this
    process touches the GPU state it needs (a subprocess, never an exec)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    args = [os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--backend', 'nccl', '--dist-at-world-1', '--steps', '1', '--warmup', '1',
            '--no-cpu-baseline', '--no-other-workloads', '--no-other-modes'] + extra
    if launcher:
        from far_amd import parallel
        cmd = parallel.launch_command(1, args)
    else:
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
def test_rccl_comes_up_at_world_size_1_eval_step():
    """init_process_group('nccl', device_id=cuda:0) under torch.distributed.run --nproc-per-node 1: RCCL initialises, the fence's
    barrier and the max-over-ranks / per-rank-gather all-reduces run on device tensors through it (mp3d_loftr/train.py:337-359
    is the deployment this stands in for; no 8-GPU node is available to the tests)."""
    res = _run_rccl_world1(['--pairs', '2', '--hyp', '256'], launcher=True)
    assert res['process_group'] == {'backend': 'nccl', 'world_size': 1}
    assert res['n_gpus'] == 1 and len(res['per_rank_ms_per_step']) == 1 and res['value'] > 0
    assert res['config']['solver_success_frac'] > 0.5


@pytest.mark.timeout(1800)
def test_rccl_world_size_1_training_step_under_ddp_and_syncbn():
    """The training workload over RCCL at world size 1: SyncBatchNorm conversion (left to torch by far_amd.ops.bn_act_train's
    predicate), DistributedDataParallel's bucketed all-reduce of the 204 MB of gradients (to itself), the stand-alone all-reduce
    and the no_sync() leg of the `exchange` block."""
    res = _run_rccl_world1(['--workload', 'c3', '--pairs', '1', '--hyp', '256'], launcher=False)
    pg = res['process_group']
    assert pg['backend'] == 'nccl' and pg['world_size'] == 1 and pg['sync_batchnorm'] is True
    assert res['config']['parallelism'].startswith('ddp1') and 'RCCL' in res['config']['parallelism']
    ex = res['exchange']
    assert ex['backend'] == 'nccl' and ex['ranks'] == 1 and ex['bytes'] > 200e6 and ex['standalone_ms'] >= 0
    assert all(v == v and abs(v) < 1e4 for v in res['config']['losses'].values())
