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


def _run(extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--share-gpu', '--steps', '1',
           '--warmup', '1', '--no-cpu-baseline'] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
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
