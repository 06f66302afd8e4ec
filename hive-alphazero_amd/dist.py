"""Multi-GPU plumbing: one process per GPU, units (boards / games) sharded by rank, no data-path
collective -- only a barrier and scalar reductions of timings/counters."""
import os

import torch
import torch.distributed as dist


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None):
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def shard(total, rank, world):
    """Contiguous shard [lo, hi) of `total` units owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def game_id_stream(rank, world, total=None):
    """Global game indices rank `rank` of `world` plays.

    total given: the contiguous shard [rank * total / world, (rank + 1) * total / world) of games 0 .. total-1, so the
    set of games a run plays -- and, the noise being keyed on the game index, every one of their records -- is the
    same for any number of GPUs.  total None (open-ended runs, bench.py): rank, rank + world, rank + 2 world, ..."""
    import itertools
    if total is None:
        return itertools.count(rank, world)
    lo, hi = shard(total, rank, world)
    return iter(range(lo, hi))


def _reduce(x, op, device):
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=op)
    return float(t.item())


def max_over_ranks(x, device="cpu"):
    return _reduce(x, dist.ReduceOp.MAX, device)


def sum_over_ranks(x, device="cpu"):
    return _reduce(x, dist.ReduceOp.SUM, device)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
