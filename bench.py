#!/usr/bin/env python3
"""bench.py -- scanned rows/s + fraction of the HBM roofline of the RangeFilter hot path.

Workload at every N (weak scaling): each rank owns 100 M-row DENSE_INT segments resident in ITS GPU's HBM
(BASELINE.json: "100M-row RangeFilter"; SURVEY 8d config C2 at 100 M rows: int32 uniform in [0, 2^30) from
splitmix64, predicate GT(2^28) AND LT(3*2^28), ~50 % selectivity).  One STEP = one pass of the hot path
ScanOp -> SelectOp(GT) -> SelectOp(LT) over one segment: the fused scan+select kernel producing the
selection bitmap (12.5 MB) and the selected-row count (reduced inside the same kernel).  When N > 1 every step's
count is logged on the device by that kernel (imm3_query_log_counts) and the K counts are summed over the ranks by
ONE RCCL all-reduce at the end of the timed region, inside it -- the only collective on the path.
Three distinct segments per rank are rotated so the 256 MiB Infinity Cache cannot serve the reads.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  value = rows scanned by all ranks / max-over-ranks wall time, inputs
already resident in HBM.  roofline.achieved = algorithmic bytes per launch (4.125 B/row: 4 B column read +
1/8 B bitmap write) / mean duration of the scan+select kernel measured live with HIP events on the
launching stream.  cpu_baseline = the oracle's faithful C restatement of the reference CPU operators
(kind "port": the Scala/JVM reference cannot run here), one thread -- the reference runs one thread per
segment (Engine.scala:176-180) -- timed on rank 0 at N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
ROWS_PER_SEGMENT = 100_000_000
ALGO_BYTES_PER_ROW = 4.125     # 4 B int32 read + 1/8 B bitmap write


class _DevArray:
    """Zero-copy view of library-owned device memory for torch (cuda array interface)."""

    def __init__(self, ptr: int, n: int, typestr: str = "<i8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 3}


def cpu_baseline(values: np.ndarray, offsets: np.ndarray, sels, budget_s: float = 10.0):
    """The oracle (kind 'port': the Scala/JVM reference cannot run here), on the same 100 M-row segment.
    Primary figure: faithful flavour (the reference's per-block copy / per-element decode / per-row BitSet cost
    structure), ONE thread -- the reference runs one thread per segment (Engine.scala:176-180).  Also reported,
    labelled: the tight flavour on one thread, and an all-cores run with the rows split README-style into 98
    segments (README.md:10: block 1024 x segment 1000), one thread per segment up to the CPUs this process may use."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_c
    n = values.shape[0]
    raw = values.view(np.uint8)
    col = oracle_c.OColumn(raw, offsets, oracle_c.DENSE_INT, 4)

    def timed(flavour, max_reps):
        t0 = time.perf_counter()
        reps, count = 0, 0
        while True:
            _, count = oracle_c.scan_select([col], sels, 1024, flavour)
            reps += 1
            if time.perf_counter() - t0 >= budget_s or reps >= max_reps:
                break
        return n * reps / (time.perf_counter() - t0), reps, count

    faithful, reps, count = timed(0, 8)
    tight, _, _ = timed(1, 3)
    # all cores: 98 segments (1 024 000 rows each, last one shorter); ctypes releases the GIL inside the C call
    seg_rows = 1024 * 1000
    bounds = [(s, min(s + seg_rows, n)) for s in range(0, n, seg_rows)]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(cores, len(bounds)))
    seg_cols = [oracle_c.OColumn(raw[4 * a: 4 * b], offsets[: (b - a + 1023) // 1024 + 1].copy() if (b - a) % 1024 == 0 else
                                 np.concatenate([offsets[: (b - a) // 1024 + 1], [4 * (b - a)]]).astype(np.int32),
                                 oracle_c.DENSE_INT, 4) for a, b in bounds]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        counts = list(ex.map(lambda c: oracle_c.scan_select([c], sels, 1024, 0)[1], seg_cols))
    all_cores = n / (time.perf_counter() - t0)
    assert sum(counts) == count
    return {
        "value": faithful,
        "unit": "rows/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{reps} full passes over one {n}-row DENSE_INT segment (same data and predicate as the GPU step), "
                  f"oracle/imm3_oracle.c faithful flavour, gcc -O2, 1 thread (host has {os.cpu_count()} logical CPUs, "
                  f"{cores} usable by this process)",
        "selected_rows": int(count),
        "tight_flavour_1_thread": tight,
        "all_cores": {"value": all_cores, "cores": workers, "segments": len(bounds),
                      "note": "faithful flavour, one thread per segment, README-style segments (1024 x 1000 rows) of the same rows"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=ROWS_PER_SEGMENT, help="rows per segment (default 100M)")
    ap.add_argument("--segments", type=int, default=3, help="distinct resident segments rotated per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra", action="store_true", help="also time C3 (range+project) and C4 (match+project)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--grid", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the immutable3 hot path has no CPU fallback")
    # Rehearsal knobs (not used by the driver): IMM3_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # IMM3_BENCH_BACKEND=gloo swaps RCCL for gloo, so the N > 1 code path can be exercised on a 1-GPU box.
    if os.environ.get("IMM3_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("IMM3_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    # IMM3_BENCH_FORCE_DIST=1 (rehearsal): take the N > 1 code path -- RCCL init, per-step count all-reduce, barriers --
    # even with one rank, so that path can be exercised on a 1-GPU box with the real backend.
    use_dist = world > 1 or os.environ.get("IMM3_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # "nccl" IS RCCL on ROCm
        else:
            dist.init_process_group(backend=backend)

    from immutable3_amd import native, synth

    n = args.rows
    sels = [(0, native.GT, float(2 ** 28)), (0, native.LT, float(3 * 2 ** 28))]
    stream = torch.cuda.current_stream().cuda_stream
    ctx = native.Context(local_rank, stream)   # launch on torch's current stream: its synchronize() covers us
    ctx.set_tuning(args.variant, args.grid)
    offsets = synth.block_offsets(n, 4)

    # ---- stage: segments resident in HBM before any timed region (PCIe staging is not part of `value`) ----
    segs, queries, host0 = [], [], None
    t_stage = time.perf_counter()
    for s in range(args.segments):
        seed = 1 + s + 1000 * rank          # rank r, slot s: its own splitmix64 stream
        v = synth.uniform_int30(seed, n)
        if s == 0 and rank == 0:
            host0 = v
        seg = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, v.view(np.uint8), n * 4, offsets)])
        segs.append(seg)
        queries.append(native.DeviceQuery(ctx, seg, [0], sels, [], 0, 1024))
        del v
    stage_s = time.perf_counter() - t_stage

    # parity gate for the reported number: popcount(bitmap) == count == numpy evaluation of segment 0
    if rank == 0:
        queries[0].run_select()
        words = queries[0].bitmap()
        cnt = queries[0].count()
        keep = (host0 > 2 ** 28) & (host0 < 3 * 2 ** 28)
        assert cnt == int(keep.sum()), (cnt, int(keep.sum()))
        assert words.tobytes() == np.packbits(keep, bitorder="little").tobytes(), "bitmap mismatch"
        del keep, words

    # N > 1: "RCCL over xGMI only for the final selected-row-count reduction".  Every scan stores its count into a
    # device-side log from the kernel that produces it (imm3_query_log_counts: no copy kernel, no host call per step);
    # the K per-step counts are summed over the ranks by ONE all-reduce at the end of the timed region, inside it.
    nq = len(queries)
    per_query = (args.steps + nq - 1) // nq + 1
    logs = [torch.zeros(per_query, dtype=torch.int64, device="cuda") for _ in queries]
    counts = torch.zeros(args.steps, dtype=torch.int64, device="cuda")

    def arm_logs():
        for q, lg in zip(queries, logs):
            lg.zero_()
            torch.cuda.synchronize()
            q.log_counts(lg.data_ptr() if use_dist else 0, per_query)

    def step(i: int):
        queries[i % nq].run_select()                     # fused ScanOp -> SelectOp(GT) -> SelectOp(LT) kernel (+ count)

    def reduce_counts():
        # step i of the timed region ran query i % nq as that query's (i // nq)-th logged run
        for j in range(nq):
            counts[j::nq] = logs[j][: len(range(j, args.steps, nq))]
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)    # final selected-row-count reduction over RCCL / xGMI

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    arm_logs()
    # start the timed steps on query 0 again so that log slot k of query j is timed step j + k * nq
    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides ----
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if use_dist:
        reduce_counts()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    expected_counts = counts.clone() if use_dist else None
    if use_dist and world == 1 and rank == 0:   # one rank: the reduced count of step 0 is segment 0's count (parity gate above)
        assert int(expected_counts[0].item()) == cnt, (int(expected_counts[0].item()), cnt)
    for q in queries:
        q.log_counts(0, 0)

    # ---- the same K steps again with the scan+select kernel bracketed by HIP events on its stream.  Kept out
    # of the region above because the event packets themselves cost ~5 us per step; the work is identical. ----
    ctx.timing_enable(args.steps + 8)
    ctx.timing_mask(1 << 0)
    ctx.timing_reset()
    ctx.devclock_enable(args.steps + 8)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    elapsed_events = time.perf_counter() - t1
    kernel_ms = ctx.timing_collect(0)
    devclock_ms = ctx.devclock_collect()
    ctx.timing_enable(0)
    ctx.devclock_enable(0)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    read_ceiling = ctx.measure_read_gbps(4 * n, 30) if rank == 0 else None   # read-only streaming kernel on this box
    if rank == 0:
        total_rows = float(n) * args.steps * world
        mean_ms = float(np.mean(kernel_ms)) if kernel_ms.size else float("nan")
        achieved = ALGO_BYTES_PER_ROW * n / (mean_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == "range_filter_i32" and tj.get("rows") == n:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "scanned rows/sec + %HBM-roofline, 100M-row RangeFilter, 1/2/4/8 MI355X",
            "value": total_rows / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_with_kernel_events": elapsed_events / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32",
            "data": "synthetic",
            "config": {
                "workload": "C2@100M: RangeFilter GT(2^28) AND LT(3*2^28) over one 100M-row DENSE_INT segment "
                            "-> selection bitmap + selected-row count",
                "rows_per_step_per_gpu": n,
                "block_rows": 1024,
                "segments_rotated_per_gpu": args.segments,
                "selectivity": 0.5,
                "parallelism": f"segment-sharded x{world}, one count all-reduce over RCCL per K steps" if world > 1 else "1 GPU",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "imm3::k_filter_tile<TK_I32>",
                "kernel_ms_mean": mean_ms,
                "kernel_ms_min": float(np.min(kernel_ms)) if kernel_ms.size else None,
                "kernel_launches_timed": int(kernel_ms.size),
                "kernel_ms_mean_device_clock": float(np.mean(devclock_ms)) if devclock_ms.size else None,
                "timing": "HIP events stamped by hipExtLaunchKernelGGL on the launching stream, second pass of the same K steps; "
                          "reads ~4 us above rocprofv3's kernel-only duration (start stamp precedes dispatch), see DESIGN.md section 6; "
                          "kernel_ms_mean_device_clock = first work-group entry to last work-group exit on the 100 MHz device clock, same launches",
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ROW * n,
                "empirical_read_ceiling_GBps": read_ceiling,
                "frac_of_empirical_read_ceiling": (achieved / read_ceiling) if read_ceiling else None,
            },
            "staging": {"host_to_hbm_s_per_segment": stage_s / args.segments, "note": "PCIe staging incl. synthetic generation; never part of value"},
        }
        if use_dist:
            result["count_allreduce"] = {"collective": "one RCCL all_reduce(SUM) over the K per-step counts at the end of the timed region",
                                         "last_step_global_count": int(expected_counts[-1].item()),
                                         "sum_over_steps": int(expected_counts.sum().item())}

    if args.extra and rank == 0 and world == 1:
        result["extra"] = extra_workloads(ctx, native, synth, n)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(host0, offsets, [(0, 3, float(2 ** 28)), (0, 4, float(3 * 2 ** 28))])
    elif rank == 0:
        result["cpu_baseline"] = None

    for q in queries:
        q.close()
    for s in segs:
        s.close()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def extra_workloads(ctx, native, synth, n, steps: int = 30):
    """C3 (conjunctive RangeFilter on age and id + Project(id, age)) and C4 (MatchFilter(state)=='CA' +
    Project(id, state, age)) over one 100 M-row segment: not bench lines, reported for the record."""
    import torch
    out = {}
    ids = np.arange(n, dtype=np.int32)
    age = synth.uniform_below(2, n, 100, np.int8)
    st = synth.state_codes(3, n)
    seg = native.DeviceSegment(ctx, [
        (native.DENSE_INT, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4)),
        (native.DENSE_STRING, 2, st.reshape(-1), n * 2, synth.block_offsets(n, 2)),
        (native.DENSE_TINYINT, 1, age.view(np.uint8), n, synth.block_offsets(n, 1)),
    ])
    cases = {
        "c3_range_age_id_project": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, 1e6), (1, native.LT, 9e7)], [1, 0],
                                    lambda sel: 5 + 0.125 + sel * (4 + 5 + 5)),
        "c4_match_state_project": ([1, 0, 2], [(0, native.MATCH, [b"CA"])], [1, 0, 2],
                                   lambda sel: 2 + 0.125 + sel * (4 + 7 + 7)),
    }
    for name, (used, sels, proj, bytes_per_row) in cases.items():
        q = native.DeviceQuery(ctx, seg, used, sels, proj, 0, 1024)
        q.run()
        cnt = q.count()
        q.reserve_rows(cnt + 1024)
        for _ in range(3):
            q.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):                       # wall time without instrumentation ...
            q.run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ctx.timing_enable(4 * steps + 8)             # ... kernel durations from a second, event-bracketed pass
        ctx.timing_mask(0xFFFFFFFF)
        ctx.timing_reset()
        for _ in range(steps):
            q.run()
        torch.cuda.synchronize()
        k = [ctx.timing_collect(i) for i in range(3)]
        ctx.timing_enable(0)
        sel = cnt / n
        out[name] = {
            "rows_per_s": n / dt, "ms_per_query": dt * 1e3, "selected_rows": int(cnt), "selectivity": sel,
            "algorithmic_bytes_per_row": bytes_per_row(sel),
            "algorithmic_GBps": bytes_per_row(sel) * n / dt / 1e9,
            "kernel_ms": {"scan_select": float(np.mean(k[0])) if k[0].size else None,
                          "offsets_scan": float(np.mean(k[1])) if k[1].size else None,
                          "compact_gather": float(np.mean(k[2])) if k[2].size else None},
        }
        q.close()
    # group-by aggregation: select count(id), max(age) from t [where age > 18 and age < 30] group by state
    for name, sels in (("agg_group_by_state_all_rows", []),
                       ("agg_group_by_state_range_age", [(1, native.GT, 18.0), (1, native.LT, 30.0)])):
        q = native.DeviceQuery(ctx, seg, [1, 2, 0], sels, (), 0, 1024, group_cols=[0], aggs=[(native.AGG_COUNT, 2), (native.AGG_MAX, 1)])
        for _ in range(2):
            q.run()
        torch.cuda.synchronize()
        ctx.timing_enable(8 * 10 + 8)
        ctx.timing_mask(0xFFFFFFFF)
        ctx.timing_reset()
        t0 = time.perf_counter()
        for _ in range(10):
            q.run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        k0, k4 = ctx.timing_collect(0), ctx.timing_collect(4)
        ctx.timing_enable(0)
        keys, first, counts, vals = q.fetch_groups()
        out[name] = {"rows_per_s": n / dt, "ms_per_query": dt * 1e3, "groups": int(keys.shape[0]), "selected_rows": int(counts.sum()),
                     "kernel_ms": {"scan_select": float(np.mean(k0)) if k0.size else None, "group_agg": float(np.mean(k4)) if k4.size else None}}
        q.close()
    seg.close()
    # host -> HBM staging of one 400 MB column from pageable memory (the step before the path)
    t0 = time.perf_counter()
    s2 = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4))])
    dt = time.perf_counter() - t0
    out["staging_400MB_pageable"] = {"seconds": dt, "GBps": n * 4 / dt / 1e9}
    s2.close()
    # PFOR_INT: the id column as PFORCodecInt.encode writes it; the range predicate is evaluated on the
    # compressed blocks (k_filter_pfor), HBM traffic = compressed bytes.  VALU-bound, not HBM-bound.
    dat, offs = native.pfor_encode_column(ids, 1024)
    t0 = time.perf_counter()
    sp = native.DeviceSegment(ctx, [(native.PFOR_INT, 4, dat, dat.size, offs)])
    stage_s = time.perf_counter() - t0
    q = native.DeviceQuery(ctx, sp, [0], [(0, native.GT, 1e6), (0, native.LT, 9e7)])
    for _ in range(3):
        q.run_select()
    cnt = q.count()
    torch.cuda.synchronize()
    ctx.timing_enable(steps + 8)
    ctx.timing_mask(1)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        q.run_select()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    k0 = ctx.timing_collect(0)
    ctx.timing_enable(0)
    out["pfor_range_id"] = {"rows_per_s": n / dt, "ms_per_query": dt * 1e3, "selected_rows": int(cnt), "compressed_bytes": int(dat.size),
                            "compression_ratio": n * 4 / dat.size, "kernel_ms": {"scan_select": float(np.mean(k0)) if k0.size else None},
                            "hbm_GBps": (dat.size + n / 8) / (float(np.mean(k0)) * 1e-3) / 1e9 if k0.size else None,
                            "staging_seconds": stage_s, "bound": "valu"}
    q.close()
    sp.close()
    return out


if __name__ == "__main__":
    main()
