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


def _agg_worker(rank, world, port, n_segments, out):
    import torch.distributed as dist
    from immutable3_amd.distributed import ShardedAggregate
    from oracle import oracle_np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def local_agg(seg):
        cols, masks = _agg_segment(seg)
        res = oracle_np.project_agg(cols, [1], AGGS, masks)
        return list(res.items())

    merged = ShardedAggregate(n_segments, rank, world, local_agg, ["count", "max", "min"]).run()
    out[rank] = list(merged.items())
    dist.barrier()
    dist.destroy_process_group()


AGGS = [("count", 0), ("max", 0), ("min", 0)]


def _agg_segment(seg, n=3000):
    from immutable3_amd import synth
    from oracle import oracle_np
    ids = synth.uniform_below(300 + seg, n, 1000, np.int32)
    grp = synth.uniform_below(400 + seg, n, 5 + seg, np.int8)          # later segments introduce new groups
    br = blocks_of(n, 1024)
    a = RawColumn(DENSE_INT, 4, ids, br)
    b = RawColumn(2, 1, grp, br)
    cols = [a.npcol(), b.npcol()]
    _, _, masks = oracle_np.scan_select(cols, [(0, GT, 100.0)], 1024)
    return cols, masks


def test_world2_sharded_aggregate():
    from oracle import oracle_np
    world, n_segments = 2, 5
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_agg_worker, args=(world, port, n_segments, out), nprocs=world, join=True)
    per_seg = []
    for seg in range(n_segments):
        cols, masks = _agg_segment(seg)
        per_seg.append(oracle_np.project_agg(cols, [1], AGGS, masks))
    expect = oracle_np.combine_agg(per_seg, AGGS)
    assert out[0] == out[1] == [(k, v) for k, v in expect.items()]


def _c5_worker(rank, world, port, out):
    """BASELINE config C5's partition: 8 segments, segment s on rank s mod G, one count all-reduce."""
    import torch.distributed as dist
    from immutable3_amd.distributed import ShardedCount, owned_segments
    from oracle import oracle_c
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def local_count(seg):
        v = _segment(seg, 3000)
        col = RawColumn(DENSE_INT, 4, v, blocks_of(v.size, 1024))
        return oracle_c.scan_select([col.ocol()], SELS, 1024)[1]

    local, total = ShardedCount(8, rank, world, local_count).run()
    out[rank] = (local, total, owned_segments(8, rank, world))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_c5_partition_8_segments_over_g_ranks(world):
    out = mp.Manager().dict()
    mp.spawn(_c5_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    each = []
    for seg in range(8):
        v = _segment(seg, 3000)
        each.append(int(((v > 2 ** 28) & (v < 3 * 2 ** 28)).sum()))
    owned = [out[r][2] for r in range(world)]
    assert sorted(s for o in owned for s in o) == list(range(8))               # every segment exactly once
    assert all(o == [s for s in range(8) if s % world == r] for r, o in enumerate(owned))
    assert all(len(o) == 8 // world for o in owned)                             # balanced for G in {1, 2, 4, 8}
    for r in range(world):
        assert out[r][0] == sum(each[s] for s in owned[r]) and out[r][1] == sum(each)
