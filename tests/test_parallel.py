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
