"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- contiguous shards, no data-path collective,
barrier + max-over-ranks timing -- with the oracle standing in for the GPU kernels."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, q):
    sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hexl_fpga_amd  # noqa: F401
    from hexl_fpga_amd.sharding import max_over_ranks, shard_range
    import orc
    from ks_util import KsCase
    total = 5
    case = KsCase(orc, 64, 2, 3, seed=4, bits=30)
    b, e = shard_range(total, world, rank)
    mine = {}
    for i in range(b, e):                       # independent ciphertexts: no communication
        t, r = case.inputs(orc, i)
        mine[i] = int(orc.fnv(case.expected(orc, t, r)))
    dist.barrier()
    slowest = max_over_ranks(0.5 + rank)        # rank 1 is "slower"
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)      # test-only: collect for the assertion
    if rank == 0:
        q.put((slowest, gathered))
    dist.destroy_process_group()


def test_shard_range_partitions():
    sys.path.insert(0, str(ROOT))
    import hexl_fpga_amd  # noqa: F401
    from hexl_fpga_amd.sharding import shard_range
    for total in (0, 1, 5, 8, 8192, 8195):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    slowest, gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert slowest == pytest.approx(1.5)                    # MAX over ranks, not rank 0's own time
    merged = {}
    for part in gathered:
        assert not (set(part) & set(merged)), "an item was processed by two ranks"
        merged.update(part)
    assert sorted(merged) == list(range(5))                 # every ciphertext exactly once
    sys.path[:0] = [str(ROOT / "oracle"), str(ROOT / "tests")]
    import orc
    from ks_util import KsCase
    case = KsCase(orc, 64, 2, 3, seed=4, bits=30)
    for i in range(5):
        t, r = case.inputs(orc, i)
        assert merged[i] == int(orc.fnv(case.expected(orc, t, r)))
