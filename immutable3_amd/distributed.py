"""Segment-sharded execution across GPUs: one process per GPU, torch.distributed (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" on CPU for tests).

The reference runs one independent pipeline per segment with no shared state but the output queue
(engine/src/main/scala/immutabledb/engine/Engine.scala:176-180, 247-262); selection bitmaps and oids are
segment-local (Scan.scala:60).  So the path shards by segment with NO data-path collective: segment s belongs to
rank s mod world.  The only exchange is the final selected-row count: one 8-byte sum all-reduce.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence


def owner_of(segIdx: int, world: int) -> int:
    """segment s -> GPU s mod G."""
    return segIdx % world


def owned_segments(n_segments: int, rank: int, world: int) -> List[int]:
    return [s for s in range(n_segments) if owner_of(s, world) == rank]


def segment_filter(rank: int, world: int) -> Callable[[str, int], bool]:
    """Filter for operators.GpuSegmentManager: this rank stages and scans only its own segments."""
    return lambda tableName, segIdx: owner_of(segIdx, world) == rank


def allreduce_count(local_count, device=None, async_op: bool = False):
    """Sum of the per-rank selected-row counts.  `local_count` is an int or a 1-element int64 tensor (e.g. a
    zero-copy view of imm3_query_device_ptr(q, 1)).  Without an initialised process group it is the identity."""
    import torch
    import torch.distributed as dist

    if isinstance(local_count, int):
        t = torch.tensor([local_count], dtype=torch.int64, device=device or "cpu")
    else:
        t = local_count
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return (t, None) if async_op else int(t.item())
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)
    if async_op:
        return t, work
    return int(t.item())


class ShardedCount:
    """`select count` over a table whose segments are sharded over the ranks: every rank runs `local_count(seg)`
    (the fused scan+select kernel through the C ABI) on its own segments; one all-reduce yields the total."""

    def __init__(self, n_segments: int, rank: int, world: int, local_count: Callable[[int], int], device=None):
        self.n_segments, self.rank, self.world = n_segments, rank, world
        self.local_count, self.device = local_count, device

    def run(self):
        mine = owned_segments(self.n_segments, self.rank, self.world)
        local = sum(int(self.local_count(s)) for s in mine)
        return local, allreduce_count(local, self.device)


def allgather_groups(local_groups):
    """Group-by partial aggregates of every rank -> every rank.  `local_groups` is a picklable list of
    (segIdx, position_in_segment, groupKey, [state per aggregate]) for the segments this rank owns.  The tables are
    tiny (one row per group per segment), so one all_gather_object is the whole exchange -- the second and last
    collective of the path (after the count all-reduce)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(local_groups)
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, list(local_groups))
    return [g for part in gathered for g in part]


def merge_groups(groups, kinds):
    """ProjectAggregateQueueOp's combine (engine/.../operator/ProjectAggregateQueue.scala:17-49) over the gathered
    partials, first arrival first with arrival order DEFINED as (segment index, first-seen position) -- the same
    result on every rank and the same as a single process visiting the segments in ascending order.
    kinds: per aggregate 'count' | 'min' | 'max' | 'maxstr'.  Returns an ordered dict key -> [states]."""
    out = {}
    for segIdx, pos, key, states in sorted(groups, key=lambda g: (g[0], g[1])):
        cur = out.get(key)
        if cur is None:
            out[key] = list(states)
            continue
        for j, kind in enumerate(kinds):
            if kind == "count":
                cur[j] += states[j]
            elif kind == "min":
                cur[j] = min(cur[j], states[j])
            elif kind == "max":
                cur[j] = max(cur[j], states[j])
            else:  # MaxStringAggr: "" means unset
                cur[j] = states[j] if cur[j] == "" or states[j] > cur[j] else cur[j]
    return out


class ShardedAggregate:
    """`select agg(..) .. group by ..` over segments sharded s mod G: every rank aggregates its own segments with
    `local_agg(seg) -> ordered [(groupKey, [states])]` (the GPU hash-aggregation kernel through the C ABI), the
    per-segment group tables are all-gathered and merged."""

    def __init__(self, n_segments: int, rank: int, world: int, local_agg, kinds):
        self.n_segments, self.rank, self.world, self.local_agg, self.kinds = n_segments, rank, world, local_agg, list(kinds)

    def run(self):
        local = []
        for seg in owned_segments(self.n_segments, self.rank, self.world):
            for pos, (key, states) in enumerate(self.local_agg(seg)):
                local.append((seg, pos, key, list(states)))
        return merge_groups(allgather_groups(local), self.kinds)
