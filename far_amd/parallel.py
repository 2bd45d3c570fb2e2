"""Data-parallel sharding of image pairs across the GPUs of one node.

Evaluation has no data-path collective: pairs are independent, rank r takes pairs r::W (what the reference's
DistributedSampler(shuffle=False) does, mp3d_loftr/src/lightning/data.py:115-117).  The only exchanges are the
timing reduction of bench.py and an optional gather of per-pair results to rank 0 (the reference gathers pickled
metrics over a gloo side group, src/utils/comm.py:141-219 -- out of scope; this is the tensor-only equivalent).
Works with backend 'nccl' (= RCCL on ROCm) and 'gloo' (CPU tests).
"""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _group_up():
    """A process group exists (also one of world size 1: its collectives still run through the backend -- bench.py
    --dist-at-world-1 executes RCCL that way on a one-GPU box)."""
    return dist.is_available() and dist.is_initialized()


def shard_indices(n_pairs, rank=None, world_size=None):
    """Indices of the pairs this rank processes: rank::world (every pair exactly once, sizes differ by <= 1)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n_pairs, world_size))


def max_over_ranks(seconds, device='cpu'):
    """bench.py's timing rule: the slowest rank defines the step time."""
    _, w = world()
    if not _group_up():
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_pair_results(local, n_pairs, dst=0):
    """local: (n_local, D) tensor of per-pair results for shard_indices(n_pairs) in that order.
    Returns the (n_pairs, D) tensor in global pair order on rank `dst`, None elsewhere."""
    rank, w = world()
    if w == 1:
        return local
    D = local.shape[1]
    cap = (n_pairs + w - 1) // w
    buf = torch.zeros(cap, D, dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(w)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    full = torch.empty(n_pairs, D, dtype=local.dtype, device=local.device)
    for r in range(w):
        idx = shard_indices(n_pairs, r, w)
        full[idx] = out[r][:len(idx)]
    return full


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(n_ranks, script_argv, port=None):
    """The one-node launch line the driver itself uses for N > 1 (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(n_ranks)}',
            '--master-addr', '127.0.0.1', '--master-port', str(port or free_port())] + list(script_argv)


def launch_ranks(n_ranks, script_argv, port=None, env=None):
    """Start `script_argv` as n_ranks child ranks under torch.distributed.run and return its exit code.

    Must be called BEFORE anything in the calling process touches the GPU: the ranks are children (never an exec of
    a process that initialised HIP), the parent only waits.  HSA_ENABLE_IPC_MODE_LEGACY=0 is kept in the children's
    environment (RCCL needs dmabuf IPC on this driver)."""
    e = dict(os.environ if env is None else env)
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    return subprocess.call(launch_command(n_ranks, script_argv, port), env=e)


def gather_floats(value, device='cpu'):
    """Every rank's scalar as a python list on every rank (per-rank step times of bench.py)."""
    _, w = world()
    if not _group_up():
        return [float(value)]
    t = torch.zeros(w, dtype=torch.float64, device=device)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]
