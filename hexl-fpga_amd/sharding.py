"""Multi-GPU partitioning of the hot path: independent ciphertexts shard across ranks, no collective
on the data path (the reference's DevicePool: one runner per board on a shared queue,
host/src/fpga.cpp:1646-1673). torch.distributed is used only for the timing barrier / max-reduce."""
from __future__ import annotations


def shard_range(total: int, world: int, rank: int) -> tuple[int, int]:
    """contiguous block [begin, end) of `total` work items owned by `rank` (sizes differ by at most 1)"""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def max_over_ranks(seconds: float, device=None, force: bool = False) -> float:
    """wall time of the slowest rank (bench.py contract); identity when not distributed. `force`: run the reduce even in a
    group of one rank (bench.py's HEXL_BENCH_FORCE_DIST: the RCCL path executed on a one-GPU box)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
