"""N > 1 path on CPU: world_size-2 gloo processes, segments sharded s mod G, 8-byte count all-reduce.
The per-segment executor is the oracle here (no GPU in this suite); on the GPU box bench.py drives the same
sharding + all-reduce with the HIP kernel as the executor."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import DENSE_INT, GT, LT, RawColumn, blocks_of


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _segment(seg, n=5000):
    from immutable3_amd import synth
    return synth.uniform_int30(100 + seg, n)


SELS = [(0, GT, float(2 ** 28)), (0, LT, float(3 * 2 ** 28))]


def _worker(rank, world, port, n_segments, out):
    import torch
    import torch.distributed as dist
    from immutable3_amd.distributed import ShardedCount, owned_segments
    from oracle import oracle_c
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def local_count(seg):
        v = _segment(seg)
        col = RawColumn(DENSE_INT, 4, v, blocks_of(v.size, 1024))
        return oracle_c.scan_select([col.ocol()], SELS, 1024)[1]

    local, total = ShardedCount(n_segments, rank, world, local_count).run()
    # async flavour used by bench.py: the all-reduce overlaps the next scan
    from immutable3_amd.distributed import allreduce_count
    t, work = allreduce_count(torch.tensor([local], dtype=torch.int64), async_op=True)
    work.wait()
    out[rank] = (local, total, int(t.item()), owned_segments(n_segments, rank, world))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_segments", [8, 5])
def test_world2_sharded_count(n_segments):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_segments, out), nprocs=world, join=True)
    expect_each = []
    for seg in range(n_segments):
        v = _segment(seg)
        expect_each.append(int(((v > 2 ** 28) & (v < 3 * 2 ** 28)).sum()))
    assert sorted(out[0][3] + out[1][3]) == list(range(n_segments))
    assert out[0][3] == [s for s in range(n_segments) if s % 2 == 0]
    for r in range(world):
        local, total, total_async, mine = out[r]
        assert local == sum(expect_each[s] for s in mine)
        assert total == total_async == sum(expect_each)


def test_single_process_identity():
    from immutable3_amd.distributed import ShardedCount, allreduce_count, owner_of
    assert allreduce_count(41) == 41
    assert [owner_of(s, 8) for s in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    local, total = ShardedCount(3, 0, 1, lambda s: s + 1).run()
    assert local == total == 6
