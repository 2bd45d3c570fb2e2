"""World-size-2 gloo tests (CPU) of the N > 1 host logic: sharding, timing reduction, result gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from far_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        idx = parallel.shard_indices(n_pairs)
        local = torch.tensor([[float(i), float(i) * 2 + rank * 0] for i in idx], dtype=torch.float64).reshape(-1, 2)
        full = parallel.gather_pair_results(local, n_pairs)
        t = parallel.max_over_ranks(1.0 + rank)
        dist.barrier()
        q.put((rank, idx, None if full is None else full.tolist(), t))
    finally:
        dist.destroy_process_group()


def test_shard_indices_partition():
    for n in [0, 1, 7, 32, 33]:
        for w in [1, 2, 4, 8]:
            got = sorted(i for r in range(w) for i in parallel.shard_indices(n, r, w))
            assert got == list(range(n))
            sizes = [len(parallel.shard_indices(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_gather_and_timing():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    n_pairs = 7
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, idx0, full0, t0), (r1, idx1, full1, t1) = res
    assert idx0 == [0, 2, 4, 6] and idx1 == [1, 3, 5]
    assert full1 is None
    assert full0 == [[float(i), float(2 * i)] for i in range(n_pairs)]
    assert t0 == t1 == 2.0                      # slowest rank


def test_bench_launches_n_ranks_when_started_plainly():
    """`python bench.py --gpus 2` without WORLD_SIZE in the environment must start 2 ranks itself (the round-1 bench
    parsed --gpus and ignored it).  The self-test mode joins a gloo group and does the timing reduction only."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--selftest-launcher',
                          '--backend', 'gloo'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout                # rank 0 prints ONE JSON line
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['requested_gpus'] == 2
    assert res['per_rank_s'] == [0.001, 0.002] and res['max_s'] == 0.002


def test_launch_command_is_the_drivers_line():
    cmd = parallel.launch_command(8, ['bench.py', '--gpus', '8'], port=29511)
    assert cmd[1:] == ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=8', '--master-addr', '127.0.0.1',
                       '--master-port', '29511', 'bench.py', '--gpus', '8']
