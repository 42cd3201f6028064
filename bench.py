#!/usr/bin/env python3
"""bench.py -- scanned rows/s + fraction of the HBM roofline of the scan / filter / project hot path.

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # or is started as a rank

N = 1 (headline; BASELINE.json "100M-row RangeFilter", SURVEY 8d config C2 at 100 M rows): three 100 M-row DENSE_INT
segments resident in HBM (int32 uniform in [0, 2^30) from splitmix64, rotated so the 256 MiB Infinity Cache cannot
serve the reads), predicate GT(2^28) AND LT(3*2^28), ~50 % selectivity.  One STEP = one pass of ScanOp -> SelectOp(GT)
-> SelectOp(LT) over one segment: the fused scan+select kernel producing the selection bitmap (12.5 MB) and the
selected-row count (reduced inside the same kernel).  The line also carries an `extra` block, measured in the same
run: C3 (range on age and id + Project), C4 (Match(state) + Project), group-by aggregation, and C5 at G = 1 (extra.c5).

N > 1: `value` stays the SAME workload -- every rank runs the N = 1 step on its own segments (weak scaling: per-GPU work
fixed), value = N x 1e8 rows x K / max-over-ranks wall time -- so a scaling curve read from `value` compares like with like.
BASELINE config C5 (strong scaling) rides in `extra.c5` at every N: 8 segments x 100 M rows (segment s: age from splitmix64
seed 100+s, id = s*10^8 + i), segment s on rank s mod N (Engine.scala:176-180 fans out one pipeline per segment), query
RangeFilter(age) AND RangeFilter(id) + Project(id, age); one pass = every rank's segments followed by ONE count all-reduce:
ncclAllReduce(sum, uint64, 1) over RCCL / xGMI, issued by libimm3 (imm3_comm_allreduce_count) on the communicator's stream
behind the scans that produce the counts; extra.c5.value = 8e8 rows x K / max-over-ranks wall time.  One pass is launched as
one hipGraph (imm3_graph_launch; --no-graph: kernel by kernel).  At N > 1 extra.c5 also carries `g1_same_run` -- the G = 1
point of the curve measured in the same run: every rank runs all 8 segments alone on its own GPU, max over ranks (--no-g1
skips it) -- and `efficiency` = value / (N x g1_same_run.value).

Prints ONE JSON line on rank 0.  Inputs are resident in HBM before any timed region.  roofline.achieved = algorithmic
bytes per launch / mean kernel duration measured live with HIP events on the launching stream.  cpu_baseline = the
oracle's faithful C restatement of the reference CPU operators (kind "port": the Scala/JVM reference cannot run
here), one thread -- the reference runs one thread per segment -- timed on rank 0 at N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)
ROWS_PER_SEGMENT = 100_000_000
ALGO_BYTES_PER_ROW = 4.125     # 4 B int32 read + 1/8 B bitmap write
C5_SEGMENTS = 8
METRIC = "scanned rows/sec + %HBM-roofline, 100M-row RangeFilter, 1/2/4/8 MI355X"


HEADLINE_KERNEL = "void imm3::k_filter_tile<0, 3, 3, 1, false, true, false>(imm3::TileArgs)"   # as rocprofv3 names it
HEADLINE_SOURCES = ("immutable3_amd/csrc/imm3_kernels.hip", "immutable3_amd/csrc/imm3_tile.h", "immutable3_amd/csrc/imm3_device.h",
                    "immutable3_amd/csrc/imm3_internal.h")


def headline_source_sha16() -> str:
    """Hash of the sources the headline kernel is compiled from: profiles/traffic.json carries the one it was measured on, and a
    counter figure taken on other sources is not this kernel's."""
    import hashlib
    h = hashlib.sha256()
    for rel in HEADLINE_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


EXTRA_SOURCES = ("immutable3_amd/csrc/imm3_kernels.hip", "immutable3_amd/csrc/imm3_project.hip", "immutable3_amd/csrc/imm3_project_table.hip",
                 "immutable3_amd/csrc/imm3_agg.hip", "immutable3_amd/csrc/imm3_tile.h", "immutable3_amd/csrc/imm3_device.h", "immutable3_amd/csrc/imm3_internal.h")


def extra_source_sha16() -> str:
    """Hash of the kernel sources behind the extra block's configs (C3, C4, C5, the README-shaped table): the stamp of their
    entries in profiles/traffic.json."""
    import hashlib
    h = hashlib.sha256()
    for rel in EXTRA_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def stamped_traffic(name: str) -> dict:
    """HBM bytes of one config from profiles/traffic.json["extra"][name] (2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc
    passes of this command, summed over the config's kernels; tools/summarize_prof5.py writes it) -- only when the entry was
    measured on these kernel sources.  {} / {"source": why not} otherwise: a counter figure is never re-used across builds."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        e = json.load(open(path)).get("extra", {}).get(name)
    except Exception:
        return {"source": "profiles/traffic.json missing or unreadable"}
    if not e:
        return {"source": f"profiles/traffic.json has no entry '{name}'"}
    if e.get("source_sha16") != extra_source_sha16():
        return {"source": f"profiles/traffic.json['{name}'] REFUSED: measured on sources {e.get('source_sha16')}, this build is {extra_source_sha16()} -- re-run the --pmc passes (tools/gpu/r5_profile.sh)"}
    return dict(e, source=f"from profiles/traffic.json['{name}'] (tag {e.get('tag')}): 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes of this command over "
                          f"kernels {e.get('kernels')}, on these kernel sources (hash matches), committed; NOT re-measured in this run")


def c3_bytes_per_row(sel: float) -> float:
    """(4+1) predicate columns + 1/8 bitmap + sigma x [4 B index + (4+1) gathered + (4+1) written]."""
    return 5 + 0.125 + sel * (4 + 5 + 5)


def c4_bytes_per_row(sel: float) -> float:
    """2 (state) + 1/8 + sigma x [4 + 7 + 7]."""
    return 2 + 0.125 + sel * (4 + 7 + 7)


# =====================================================================================================================
# launcher: `python bench.py --gpus N` with N > 1 and no rank environment starts the N ranks itself
# =====================================================================================================================
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int) -> int:
    """One child process per GPU, started BEFORE this process touches the GPU (it never does: no torch import, no HIP
    call).  Rank 0's stdout (the JSON line) is relayed; any rank failing fails the run and ends the others."""
    import tempfile
    port = _free_port()
    procs = []
    with tempfile.TemporaryFile(mode="w+") as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                          stdout=out0 if r == 0 else sys.stderr, stderr=sys.stderr))
        failed = None
        while failed is None and any(p.poll() is None for p in procs):
            time.sleep(0.2)
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = (r, p.returncode)
        if failed is not None:                          # a dead rank leaves the others waiting in a barrier: end exactly the children we started
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            print(f"bench.py: rank {failed[0]} exited with code {failed[1]}; run aborted", file=sys.stderr)
            return 1
        out0.seek(0)
        text = out0.read()
    sys.stdout.write(text)
    sys.stdout.flush()
    if not any(l.startswith("{") for l in text.splitlines()):
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    return 0


# =====================================================================================================================
# CPU baseline (oracle; rank 0, N = 1 only)
# =====================================================================================================================
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


def _ocols(cols):
    from oracle import oracle_c
    return [oracle_c.OColumn(np.ascontiguousarray(v).view(np.uint8).reshape(-1), offs, codec, width) for (codec, width, v, offs) in cols]


def _select_project(ocols, sels, proj, flavour):
    """ScanOp -> SelectOp* -> ProjectOp of one segment on the CPU (the oracle: imm3o_scan_select + imm3o_project); -> rows emitted."""
    from oracle import oracle_c
    words, count = oracle_c.scan_select(ocols, sels, 1024, flavour)
    n_out = oracle_c.project(ocols, proj, 0, 1024, words, cap_rows=count)[0]
    assert n_out == count, (n_out, count)
    return count


def cpu_baseline_project(cols, sels, proj, n: int, want_rows: int, budget_s: float = 10.0):
    """The reference's CPU operators for a projecting config (C3, C4) on ONE thread -- one pipeline thread per segment,
    Engine.scala:176-180 -- restated by the oracle: SelectOp chain (faithful flavour = the reference's per-block copy / per-element
    decode / per-row BitSet cost structure; tight flavour also given) then ProjectOp (Project.scala:37-80) over the whole 100 M-row
    segment, the same data and query as the GPU leg."""
    ocols = _ocols(cols)
    out = {}
    for name, flavour in (("faithful", 0), ("tight", 1)):
        t0 = time.perf_counter()
        reps = 0
        while True:
            got = _select_project(ocols, sels, proj, flavour)
            assert got == want_rows, (got, want_rows)
            reps += 1
            if time.perf_counter() - t0 >= budget_s / 2 or reps >= 2:
                break
        out[name] = (n * reps / (time.perf_counter() - t0), reps)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"value": out["faithful"][0], "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"{out['faithful'][1]} full pass(es) of imm3o_scan_select + imm3o_project over the same {n}-row segment and query as the GPU leg, "
                      f"oracle/imm3_oracle.c faithful flavour, gcc -O2, 1 thread (host: {os.cpu_count()} logical CPUs, {cores} usable)",
            "selected_rows": int(want_rows), "tight_flavour_1_thread": out["tight"][0], "nproc": os.cpu_count()}


def cpu_baseline_c5(host_cols, n: int):
    """C5 on the CPU: the reference runs one pipeline thread per segment (Engine.scala:176-180) -- 8 segments, 8 threads, each the
    oracle's faithful SelectOp chain + ProjectOp over its 100 M-row segment (ctypes releases the GIL inside the C calls)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_c
    sels = [(0, 3, 18.0), (0, 4, 30.0), (1, 3, C5_ID_LO), (1, 4, C5_ID_HI)]
    per_seg = [_ocols([(oracle_c.DENSE_TINYINT, 1, c["age"], _block_offsets(n, 1)), (oracle_c.DENSE_INT, 4, c["id"], _block_offsets(n, 4))]) for c in host_cols]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(cores, len(per_seg)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        counts = list(ex.map(lambda oc: _select_project(oc, sels, [1, 0], 0), per_seg))
    dt = time.perf_counter() - t0
    return {"value": n * len(per_seg) / dt, "unit": "rows/s", "cores": workers, "kind": "port",
            "sample": f"one full pass of imm3o_scan_select + imm3o_project over all {len(per_seg)} segments of {n} rows (the GPU leg's data and query), one thread per "
                      f"segment as Engine.scala:176-180 runs them ({workers} threads; host: {os.cpu_count()} logical CPUs, {cores} usable), faithful flavour, gcc -O2",
            "selected_rows": int(sum(counts)), "seconds": dt, "nproc": os.cpu_count()}


def _block_offsets(n: int, width: int):
    from immutable3_amd import synth
    return synth.block_offsets(n, width)


# =====================================================================================================================
# per-rank environment
# =====================================================================================================================
class Env:
    """What every measurement needs: the rank's context (on a torch stream, so the library's work and torch's sit on ONE
    stream), host-side control collectives (gloo: barrier, max, object exchange -- never on the data path), and the
    library's RCCL communicator for the count reduce."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the immutable3 hot path has no CPU fallback")
        # Rehearsal knob (not used by the driver): IMM3_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 so the N > 1 code
        # path can be exercised on a 1-GPU box.  RCCL refuses two ranks on one device, so the count all-reduce then goes
        # over gloo (labelled in the line).
        self.one_device = os.environ.get("IMM3_BENCH_ONE_DEVICE") == "1"
        if self.one_device:
            self.local_rank = 0
        elif self.world > torch.cuda.device_count():
            raise SystemExit(f"bench.py: {self.world} ranks but {torch.cuda.device_count()} GPUs "
                             "(IMM3_BENCH_ONE_DEVICE=1 rehearses the N > 1 path on one GPU)")
        torch.cuda.set_device(self.local_rank)
        self.use_dist = self.world > 1
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo")     # host-side control plane only
        from immutable3_amd import native, synth
        self.native, self.synth = native, synth
        self.stream = torch.cuda.Stream()               # a REAL stream: 0 (torch's default) would make the library create its own
        self.ctx = native.Context(self.local_rank, self.stream.cuda_stream)
        self.ctx.set_tuning(args.variant, args.grid)
        self.comm = None
        self.count_reduce = "none (1 GPU)"

    def make_comm(self):
        """The library's RCCL communicator (also with one rank: the G = 1 point of the C5 curve runs the same code)."""
        if self.comm is not None or (self.use_dist and self.one_device):
            if self.use_dist and self.one_device:
                self.count_reduce = "REHEARSAL: torch.distributed gloo all_reduce of host counts (ranks share one GPU; RCCL refuses that)"
            return
        native = self.native
        # RCCL prints a version banner on stdout when it initialises; stdout carries the ONE JSON line, so the banner goes to stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            uid = [native.comm_unique_id() if self.rank == 0 else None]
            if self.use_dist:
                self.dist.broadcast_object_list(uid, src=0)
            self.comm = native.Comm(self.ctx, self.world, self.rank, uid[0])
            self.sync()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        self.count_reduce = ("ncclAllReduce(sum, ncclUint64, 1) per pass, issued by libimm3 (imm3_comm_allreduce_count) on the "
                             "communicator's stream behind the scans; RCCL over xGMI")

    def sync(self):
        self.torch.cuda.synchronize()                   # device-wide: the context's stream and the communicator's

    def barrier(self):
        if self.use_dist:
            self.dist.barrier()

    def max_over_ranks(self, x: float) -> float:
        if not self.use_dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: int) -> int:
        if not self.use_dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def gather_objects(self, obj):
        if not self.use_dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def timed_steps(self, step, steps: int, warmup: int) -> float:
        """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
        for i in range(warmup):
            step(i)
        self.sync()
        self.barrier()
        self.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        self.sync()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def kernel_times(self, run, reps: int, ids=(0, 1, 2, 3), launches_per_run: int = 8):
        """Duration (ms) per kernel id and run, averaged over `reps` more runs, each launch bracketed by HIP events on the
        context's stream (kept out of the wall-clock regions: the event packets cost ~5 us per launch)."""
        ctx = self.ctx
        ctx.timing_enable(launches_per_run * reps + 16)
        ctx.timing_mask(0xFFFFFFFF)
        ctx.timing_reset()
        for _ in range(reps):
            run()
        self.sync()
        ks = {i: ctx.timing_collect(i) for i in ids}
        ctx.timing_enable(0)
        return {i: (float(np.mean(k)) * (k.size / reps) if k.size else None) for i, k in ks.items()}   # ms per run (sums multi-launch ids)

    def close(self):
        if self.comm is not None:
            self.comm.close()
        self.ctx.close()
        if self.use_dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


# kernel id 0 is the scan+select launch; for a projection planned as ONE launch (k_filter_project: the filter kernel writes the rows)
# it is the whole query, and ids 1 / 2 stay empty
KERNEL_NAMES = {0: "scan_select", 1: "offsets_scan", 2: "compact_gather", 3: "count_reduce", 4: "group_agg"}


# =====================================================================================================================
# C2: the N = 1 headline (and the weak-scaling leg at N > 1)
# =====================================================================================================================
def measure_c2(env: Env, steps: int, warmup: int, with_cpu_baseline: bool):
    torch, native, synth, ctx, args = env.torch, env.native, env.synth, env.ctx, env.args
    n = args.rows
    sels = [(0, native.GT, float(2 ** 28)), (0, native.LT, float(3 * 2 ** 28))]
    offsets = synth.block_offsets(n, 4)
    segs, queries, host0 = [], [], None
    t_stage = time.perf_counter()
    for s in range(args.segments):
        v = synth.uniform_int30(1 + s + 1000 * env.rank, n)          # rank r, slot s: its own splitmix64 stream
        if s == 0:
            host0 = v
        seg = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, v.view(np.uint8), n * 4, offsets)])
        segs.append(seg)
        queries.append(native.DeviceQuery(ctx, seg, [0], sels, [], 0, 1024))
        del v
    stage_s = time.perf_counter() - t_stage

    # parity gate for the reported number: popcount(bitmap) == count == numpy evaluation of this rank's segment 0
    queries[0].run_select()
    words, cnt = queries[0].bitmap(), queries[0].count()
    keep = (host0 > 2 ** 28) & (host0 < 3 * 2 ** 28)
    assert cnt == int(keep.sum()), (cnt, int(keep.sum()))
    assert words.tobytes() == np.packbits(keep, bitorder="little").tobytes(), "bitmap mismatch"
    del keep, words
    nq = len(queries)

    def step(i: int):
        queries[i % nq].run_select()                    # fused ScanOp -> SelectOp(GT) -> SelectOp(LT) kernel (+ count)

    elapsed = env.timed_steps(step, steps, warmup)

    # the same K steps again with the scan+select kernel bracketed by HIP events on its stream (and device-clock stamps)
    ctx.timing_enable(steps + 8)
    ctx.timing_mask(1 << 0)
    ctx.timing_reset()
    ctx.devclock_enable(steps + 8)
    env.sync()
    t1 = time.perf_counter()
    for i in range(steps):
        step(i)
    env.sync()
    elapsed_events = time.perf_counter() - t1
    kernel_ms = ctx.timing_collect(0)
    devclock_ms = ctx.devclock_collect()
    ctx.timing_enable(0)
    ctx.devclock_enable(0)
    read_ceiling = ctx.measure_read_gbps(4 * n, 30) if env.rank == 0 else None   # read-only streaming kernel on this box
    mean_ms = float(np.mean(kernel_ms)) if kernel_ms.size else float("nan")
    per_rank = env.gather_objects({"rank": env.rank, "kernel_ms_mean": mean_ms, "selected_rows_segment0": int(cnt)})

    out = None
    if env.rank == 0:
        achieved = ALGO_BYTES_PER_ROW * n / (mean_ms * 1e-3) / 1e9
        traffic, tsrc = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == "range_filter_i32" and tj.get("rows") == n:
                    if tj.get("kernel") == HEADLINE_KERNEL and tj.get("source_sha16") == headline_source_sha16():
                        traffic = tj.get("hbm_bytes_per_launch")
                        tsrc = ("from profiles/traffic.json: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes of this "
                                "command on these kernel sources (kernel name and source hash match), committed; NOT re-measured in this run")
                    else:
                        tsrc = ("profiles/traffic.json REFUSED: it was measured on kernel '%s' / sources %s, this build is '%s' / %s -- "
                                "re-run the --pmc passes (tools/summarize_prof3.py)" % (tj.get("kernel"), tj.get("source_sha16"), HEADLINE_KERNEL, headline_source_sha16()))
            except Exception:
                traffic = None
        out = {
            "value": float(n) * steps * env.world / elapsed,
            "ms_per_step": elapsed / steps * 1e3,
            "ms_per_step_with_kernel_events": elapsed_events / steps * 1e3,
            "config": {
                "workload": "C2@100M: RangeFilter GT(2^28) AND LT(3*2^28) over one 100M-row DENSE_INT segment "
                            "-> selection bitmap + selected-row count",
                "rows_per_step_per_gpu": n, "block_rows": 1024, "segments_rotated_per_gpu": args.segments, "selectivity": 0.5,
                "parallelism": "1 GPU" if env.world == 1 else f"every rank scans its own segments (weak scaling) x{env.world}",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": tsrc,
                "kernel": "imm3::k_filter_tile<TK_I32>",
                "kernel_ms_mean": mean_ms,
                "kernel_ms_min": float(np.min(kernel_ms)) if kernel_ms.size else None,
                "kernel_launches_timed": int(kernel_ms.size),
                "kernel_ms_mean_device_clock": float(np.mean(devclock_ms)) if devclock_ms.size else None,
                "timing": "HIP events stamped by hipExtLaunchKernelGGL on the launching stream, second pass of the same K steps; "
                          "reads ~4 us above rocprofv3's kernel-only duration under the profiler (start stamp precedes dispatch), see DESIGN.md section 6; "
                          "kernel_ms_mean_device_clock = first work-group entry to last work-group exit on the 100 MHz device clock, same launches",
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ROW * n,
                "empirical_read_ceiling_GBps": read_ceiling,
                "frac_of_empirical_read_ceiling": (achieved / read_ceiling) if read_ceiling else None,
            },
            "staging": {"host_to_hbm_s_per_segment": stage_s / args.segments, "note": "PCIe staging incl. synthetic generation; never part of value"},
            "per_rank": per_rank,
        }
        if with_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host0, offsets, [(0, 3, float(2 ** 28)), (0, 4, float(3 * 2 ** 28))])
    for q in queries:
        q.close()
    for s in segs:
        s.close()
    return out


# =====================================================================================================================
# C5: 8 segments x 100 M rows sharded s mod G, RangeFilter(age) AND RangeFilter(id) + Project(id, age), count all-reduce
# =====================================================================================================================
C5_ID_LO, C5_ID_HI = 1.0e6, 7.9e8        # the C3 id range stretched over the 8-segment id space [0, 8e8)


def measure_c5(env: Env, steps: int, warmup: int, use_graph: bool = True, solo: bool = False, with_cpu_baseline: bool = False):
    """solo: EVERY rank runs the whole job (all 8 segments) alone on its own GPU -- the G = 1 point of the strong-scaling curve
    measured inside an N > 1 run (time = max over ranks); no count collective, the counts are checked locally.

    The segments a rank owns are ONE scan unit (imm3_table, round 5): one table query -- one launch of k_filter_project's table
    instance per pass whatever the number of owned segments (Engine.scala:176-196 merges the per-segment pipelines into one
    result); --c5-per-segment keeps round 4's one query (one launch) per segment for A/B runs."""
    torch, native, synth, ctx, args = env.torch, env.native, env.synth, env.ctx, env.args
    from immutable3_amd.distributed import owned_segments
    n = args.rows
    as_table = not args.c5_per_segment
    if not solo:
        env.make_comm()
    comm = None if solo else env.comm
    mine = list(range(C5_SEGMENTS)) if solo else owned_segments(C5_SEGMENTS, env.rank, env.world)
    sels = [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, C5_ID_LO), (1, native.LT, C5_ID_HI)]
    segs, queries, expect, host_cols = [], [], [], []
    exp_seg, exp_row, exp_id, exp_age = [], [], [], []
    t_stage = time.perf_counter()
    for k, s in enumerate(mine):
        c = synth.c3_segment(n, seed=100 + s, id_base=s * n)
        seg = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, c["id"].view(np.uint8), n * 4, synth.block_offsets(n, 4)),
                                         (native.DENSE_TINYINT, 1, c["age"].view(np.uint8), n, synth.block_offsets(n, 1))])
        keep = (c["age"] > 18) & (c["age"] < 30) & (c["id"] > C5_ID_LO) & (c["id"] < C5_ID_HI)
        want = int(keep.sum())
        rows = np.flatnonzero(keep)
        if as_table:                                    # gated below, once the table query exists
            exp_seg.append(np.full(rows.size, k, np.uint32))
            exp_row.append(rows.astype(np.uint32))
            exp_id.append(c["id"][rows])
            exp_age.append(c["age"][rows])
        else:
            q = native.DeviceQuery(ctx, seg, [1, 0], sels, [1, 0], 0, 1024)     # used columns [age, id] (Engine.getColumns), SELECT id, age
            # parity gate: count, ordered rows and values of every owned segment against numpy
            q.run()
            assert q.count() == want, (s, q.count(), want)
            q.reserve_rows(want + 1024)
            q.run()
            idx, vals = q.fetch_rows()
            assert idx.size == want and (idx == rows).all(), f"segment {s}: row order"
            assert (vals[0].view("<i4").reshape(-1) == c["id"][rows]).all() and (vals[1].view(np.int8).reshape(-1) == c["age"][rows]).all(), f"segment {s}: values"
            queries.append(q)
            del idx, vals
        if with_cpu_baseline:
            host_cols.append(c)
        expect.append(want)
        segs.append(seg)
        del c, keep, rows
    table = None
    if as_table and segs:
        table = native.DeviceTable(ctx, segs)
        q = native.DeviceQuery(ctx, table, [1, 0], sels, [1, 0], 0, 1024)
        # parity gate: count, rows in (segment, row) order and values of every owned segment against numpy
        q.run()
        assert q.count() == sum(expect), (q.count(), sum(expect))
        q.reserve_rows(sum(expect) + 1024)
        q.run()
        idx, vals = q.fetch_rows()
        seg_of, row_of = q.locate_rows(idx)
        assert idx.size == sum(expect), (idx.size, sum(expect))
        assert (seg_of == np.concatenate(exp_seg)).all() and (row_of == np.concatenate(exp_row)).all(), "table query: row order"
        assert (vals[0].view("<i4").reshape(-1) == np.concatenate(exp_id)).all() and (vals[1].view(np.int8).reshape(-1) == np.concatenate(exp_age)).all(), "table query: values"
        queries.append(q)
        del idx, vals, seg_of, row_of
    del exp_seg, exp_row, exp_id, exp_age
    stage_s = time.perf_counter() - t_stage
    expect_total = sum(expect) if solo else env.sum_over_ranks(sum(expect))     # over gloo: independent of the collective under test

    log = torch.zeros(max(steps, warmup, 1), dtype=torch.int64, device="cuda")
    env.sync()
    host_counts = []

    plan = queries[0].plan() if queries else {}
    kernels_per_query = 1 if plan.get("single_pass") else 3
    # One pass = the runs of this rank's queries (one table query; or one per owned segment).  They are recorded once
    # (imm3_ctx_capture_begin / _end: a hipGraph) and replayed with one call per pass; `--no-graph` issues the runs one by one.
    graph = None
    graph_ms = None
    # A pass that is ONE launch (the table query's one-launch plan) is issued directly: a one-kernel hipGraph costs ~30 us per launch
    # more than the kernel launch it wraps (measured in this function, round 5: 0.884-0.902 ms per pass through the graph, 0.854-0.870
    # kernel by kernel, the same launch) -- a fifth of the pass for a rank that owns one segment.  The graph is kept for passes of
    # several kernels (one query per segment: 8-24 launches), where it is what saves the time; for the record the one-launch pass is
    # timed through a graph as well (`ms_per_step_graph`).
    record_only = use_graph and kernels_per_query * len(queries) <= 2
    if use_graph:
        with ctx.capture() as cap:
            for q in queries:
                q.run()
        graph = cap.graph

    def step(i: int, graphed: bool = True):
        if graph is not None and graphed:
            graph.launch()
        else:
            for q in queries:
                q.run()                                 # scan + select + project of every owned segment
        if comm is not None:
            comm.allreduce_count(queries, device_out=log.data_ptr() + 8 * i, wait=False)
        elif solo:                                      # no collective: the counts are summed once the loop is over
            pass
        else:                                           # rehearsal on one device: host counts over gloo
            host_counts.append(env.sum_over_ranks(sum(q.count() for q in queries)))

    if record_only:                                    # (the one-launch pass through a graph, for the record; the measurement below issues it directly)
        graph_ms = env.timed_steps(step, steps, min(warmup, 2)) / steps * 1e3
        graph = None
        host_counts.clear()
    elapsed = env.timed_steps(step, steps, warmup)
    got = log[:steps].tolist() if comm is not None else ([sum(q.count() for q in queries)] if solo else host_counts[-steps:])
    assert all(g == expect_total for g in got), (got[:4], expect_total)
    # How far behind the pass's last scan the collective ends: the all-reduce sits on the communicator's stream behind the scans, the
    # next pass's scans do not wait for it -- on the host.  On the device a one-launch pass wants every CU; what a co-resident kernel
    # of the communicator costs it is measured by tools/overlap_probe.py (profiles/r05_overlap.txt).  Here: the wall time of K passes
    # with the collective against K passes without it, per pass.
    allreduce_gap_ms = None
    if comm is not None:
        def step_no_collective(i: int):
            if graph is not None:
                graph.launch()
            else:
                for q in queries:
                    q.run()
        bare = env.timed_steps(step_no_collective, steps, min(warmup, 2))
        allreduce_gap_ms = (elapsed - bare) / steps * 1e3
    elapsed_plain = None
    if graph is not None:                               # the same passes launched kernel by kernel, for the record
        elapsed_plain = env.timed_steps(lambda i: step(i, graphed=False), steps, min(warmup, 2))
        got = log[:steps].tolist() if comm is not None else ([sum(q.count() for q in queries)] if solo else host_counts[-steps:])
        assert all(g == expect_total for g in got), (got[:4], expect_total)

    # per-kernel durations of one pass over this rank's segments (second, event-bracketed pass)
    def one_pass():
        for q in queries:
            q.run()
    k = env.kernel_times(one_pass, max(3, min(steps, 20)), launches_per_run=8 * max(len(queries), 1))
    per_pass = {KERNEL_NAMES[i]: v for i, v in k.items()}
    per_query = {name: (v / len(mine) if v is not None else None) for name, v in per_pass.items()}   # per owned SEGMENT
    sel = sum(expect) / (n * len(mine))
    kernel_ms_per_query = sum(v for v in per_query.values() if v)
    # What became of the one-launch runs on this rank: a getter reads the last run's status word (bit 1: a look-back wait ran into its
    # poll cap -- not every work-group resident? -- and the rows came from the bitmap instead; bit 2: another launch of the kernel
    # owned the device).  Non-zero here on ANY rank means that rank's timings are those of the fallback, not of the plan.
    for q in queries:
        q.row_count()
    plans = [q.plan() for q in queries]
    per_rank = env.gather_objects({"rank": env.rank, "segments": mine, "selected_rows": sum(expect), "kernel_ms_per_segment": per_query,
                                   "kernel_ms_per_pass": per_pass, "queries": len(queries),
                                   "single_pass_queries": sum(1 for p in plans if p.get("single_pass")),
                                   "abandoned_runs": sum(int(p.get("abandoned_runs", 0)) for p in plans),
                                   "busy_runs": sum(int(p.get("busy_runs", 0)) for p in plans)})

    out = None
    if env.rank == 0:
        algo = c3_bytes_per_row(sel) * n
        achieved = algo / (kernel_ms_per_query * 1e-3) / 1e9
        traffic = stamped_traffic("c5_table" if as_table else "c5_per_segment")
        t_bytes = traffic.get("hbm_bytes_per_pass") if traffic.get("hbm_bytes_per_pass") and len(mine) == C5_SEGMENTS else None
        pass_ms = kernel_ms_per_query * len(mine)
        out = {
            "value": float(n) * C5_SEGMENTS * steps / elapsed,
            "ms_per_step": elapsed / steps * 1e3,
            "launch": (f"one hipGraph launch per pass ({kernels_per_query * len(queries)} kernel(s), imm3_graph_launch) + the count all-reduce"
                       if graph is not None else
                       ("the pass's one launch issued directly (imm3_query_run) + the count all-reduce" if record_only else "kernel by kernel (--no-graph)")),
            "ms_per_step_kernel_by_kernel": elapsed_plain / steps * 1e3 if elapsed_plain is not None else None,
            "ms_per_step_graph": graph_ms,   # (a one-launch pass replayed as a one-kernel hipGraph: not what `value` is measured on)
            "allreduce_gap_ms_per_pass": allreduce_gap_ms,
            "allreduce_gap_note": "wall time per pass with the count all-reduce minus without it (same K passes, same graph): what the collective adds "
                                  "behind the last scan of a pass, net of what the next pass's scans hide",
            "global_selected_rows_per_pass": int(expect_total),
            "count_allreduce": ({"collective": "none (solo: every rank runs all 8 segments on its own GPU)", "checked": f"local count == {expect_total} (numpy)"} if solo else
                                {"collective": env.count_reduce, "checked": f"all {steps} per-pass global counts == {expect_total} (numpy, summed over ranks via gloo)"}),
            "config": {
                "workload": "C5: 8 x 100M-row segments (age seed 100+s, id = s*1e8 + i), segment s -> rank s mod G, "
                            "RangeFilter(age > 18 AND age < 30) AND RangeFilter(id > 1e6 AND id < 7.9e8) + Project(id, age), "
                            "one count all-reduce per pass",
                "rows_per_step": n * C5_SEGMENTS, "segments": C5_SEGMENTS, "segments_per_gpu": len(mine), "block_rows": 1024,
                "queries_per_gpu": len(queries),
                "scan_unit": ("the rank's segments as ONE table query (imm3_table): one launch per pass" if as_table else "one query (one launch) per owned segment"),
                "selectivity": sel, "parallelism": f"segment-sharded x{env.world} (s mod G), no data-path collective but the count",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": t_bytes, "traffic_source": traffic.get("source"),
                "traffic_ratio": (t_bytes / (algo * len(mine))) if t_bytes else None,
                "frac_traffic": (t_bytes / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_bytes and pass_ms else None,
                "kernel": (("one table query per rank: ONE launch of imm3::k_filter_project's table instance over all owned segments (scan + select + project; rank 0)"
                            if as_table else "per-segment query: ONE launch, imm3::k_filter_project (scan + select + project; rank 0)") if plan.get("single_pass") else
                           "scan+select -> offsets scan -> compact+gather (rank 0)"),
                "plan": plan,
                "kernel_ms_per_pass": per_pass,
                "kernel_ms_per_query": per_query, "kernel_ms_sum_per_query": kernel_ms_per_query,
                "algorithmic_bytes_per_query": algo,
                "algorithmic_bytes_per_row": c3_bytes_per_row(sel),
                "timing": "HIP events on the launching stream, second pass; per query = per owned 100M-row segment (the pass's kernel time / segments); "
                          "frac = SURVEY 8d C3 bytes/row x rows / kernel time",
            },
            "staging": {"host_to_hbm_s_per_segment": stage_s / max(len(mine), 1), "note": "incl. synthetic generation and the parity gate; never part of value"},
            "abandoned_runs": sum(int(r.get("abandoned_runs", 0)) for r in per_rank) if per_rank else None,
            "busy_runs": sum(int(r.get("busy_runs", 0)) for r in per_rank) if per_rank else None,
            "per_rank": per_rank,
        }
        if with_cpu_baseline and host_cols:
            out["cpu_baseline"] = cpu_baseline_c5(host_cols, n)
    if graph is not None:
        graph.close()
    for q in queries:
        q.close()
    if table is not None:
        table.close()
    for s in segs:
        s.close()
    return out


# =====================================================================================================================
# extra block of the N = 1 line: C3, C4, aggregation (the other BASELINE configs at full size), staging
# =====================================================================================================================
def readme_table_c3(env: Env, ids: np.ndarray, age: np.ndarray, steps: int):
    """C3's query over the reference's own on-disk shape: `LoaderCli --block-size 1024 --segment-size 1000` (README.md:10) cuts
    100 M rows into 98 segments, each a "full" loader segment of 1000 * 1024 + 1 rows whose last block holds ONE row,
    the last one shorter.  The reference fans out one pipeline per segment and merges the rows (Engine.scala:176-196); here the 98
    resident segments are one scan unit (imm3_table) and the query is ONE launch over its tile table."""
    native, synth, ctx = env.native, env.synth, env.ctx
    n = ids.shape[0]
    seg_rows = 1000 * 1024 + 1
    bounds = [(a, min(a + seg_rows, n)) for a in range(0, n, seg_rows)]
    segs = []
    t0 = time.perf_counter()
    for a, b in bounds:
        m = b - a
        segs.append(native.DeviceSegment(ctx, [(native.DENSE_TINYINT, 1, age[a:b].view(np.uint8), m, synth.block_offsets(m, 1)),
                                               (native.DENSE_INT, 4, ids[a:b].view(np.uint8), m * 4, synth.block_offsets(m, 4))]))
    table = native.DeviceTable(ctx, segs)
    stage_s = time.perf_counter() - t0
    lo, hi = 1e6 * n / 1e8, 9e7 * n / 1e8
    sels = [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, lo), (1, native.LT, hi)]
    keep = (age > 18) & (age < 30) & (ids > lo) & (ids < hi)
    rows = np.flatnonzero(keep)
    t0 = time.perf_counter()
    q = native.DeviceQuery(ctx, table, [0, 1], sels, [1, 0], 0, 1024)
    create_ms = (time.perf_counter() - t0) * 1e3
    q.run()
    cnt = q.count()
    assert cnt == rows.size, (cnt, rows.size)
    q.reserve_rows(cnt + 1024)
    q.run()
    idx, vals = q.fetch_rows()
    seg_of, row_of = q.locate_rows(idx)
    starts = np.array([a for a, _ in bounds], dtype=np.int64)
    assert idx.size == rows.size and (starts[seg_of] + row_of == rows).all(), "README-shaped table: rows in (segment, row) order"
    assert (vals[0].view("<i4").reshape(-1) == ids[rows]).all() and (vals[1].view(np.int8).reshape(-1) == age[rows]).all(), "README-shaped table: values"
    del idx, vals, seg_of, row_of, keep
    dt = env.timed_steps(lambda i: q.run(), steps, 3) / steps
    k = env.kernel_times(q.run, steps)
    kms = {KERNEL_NAMES[i]: v for i, v in k.items()}
    ksum = sum(v for v in kms.values() if v)
    sel = cnt / n
    algo = c3_bytes_per_row(sel) * n
    q.row_count()
    tr = stamped_traffic("readme_table_c3")
    tb = tr.get("hbm_bytes_per_query")
    out = {"config": {"workload": "C3's query over 98 loader-made segments (README.md:10: block 1024 x segment 1000 -> 1 024 001 rows each, a one-row last block) "
                                  "as ONE table query (imm3_table)", "segments": len(bounds), "rows": n, "rows_per_segment": seg_rows},
           "plan": q.plan(), "create_ms": create_ms, "rows_per_s": n / dt, "ms_per_query": dt * 1e3, "selected_rows": int(cnt), "selectivity": sel,
           "algorithmic_bytes_per_row": c3_bytes_per_row(sel), "algorithmic_bytes": algo, "kernel_ms": kms, "kernel_ms_sum": ksum,
           "achieved_GBps": algo / (ksum * 1e-3) / 1e9, "frac": algo / (ksum * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_wall": algo / dt / 1e9 / HBM_PEAK_GBS,
           "roofline": {"bound": "hbm", "achieved": algo / (ksum * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / (ksum * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": tb, "traffic_source": tr.get("source"), "traffic_ratio": tb / algo if tb else None,
                        "frac_traffic": tb / (ksum * 1e-3) / 1e9 / HBM_PEAK_GBS if tb else None},
           "staging_s_98_segments": stage_s}
    q.close()
    table.close()
    for sg in segs:
        sg.close()
    return out


def extra_workloads(env: Env, steps: int = 20, with_cpu_baseline: bool = True):
    """C3 (conjunctive RangeFilter on age and id + Project(id, age)) and C4 (MatchFilter(state)=='CA' +
    Project(id, state, age)) over one 100 M-row segment, group-by aggregation, PFOR_INT: per config the algorithmic
    bytes, every kernel's mean duration (HIP events) and the fraction of the 8 TB/s peak."""
    torch, native, synth, ctx = env.torch, env.native, env.synth, env.ctx
    n = env.args.rows
    out = {}
    ids = np.arange(n, dtype=np.int32)
    age = synth.uniform_below(2, n, 100, np.int8)
    st = synth.state_codes(3, n)
    seg = native.DeviceSegment(ctx, [
        (native.DENSE_INT, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4)),
        (native.DENSE_STRING, 2, st.reshape(-1), n * 2, synth.block_offsets(n, 2)),
        (native.DENSE_TINYINT, 1, age.view(np.uint8), n, synth.block_offsets(n, 1)),
    ])
    lo, hi = 1e6 * n / 1e8, 9e7 * n / 1e8
    keep3 = (age > 18) & (age < 30) & (ids > lo) & (ids < hi)
    keep4 = (st[:, 0] == ord("C")) & (st[:, 1] == ord("A"))
    cases = {
        "c3_range_age_id_project": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, lo), (1, native.LT, hi)], [1, 0],
                                    c3_bytes_per_row, keep3),
        "c4_match_state_project": ([1, 0, 2], [(0, native.MATCH, [b"CA"])], [1, 0, 2], c4_bytes_per_row, keep4),
    }
    host_cols = [(native.DENSE_INT, 4, ids, synth.block_offsets(n, 4)), (native.DENSE_STRING, 2, st, synth.block_offsets(n, 2)),
                 (native.DENSE_TINYINT, 1, age, synth.block_offsets(n, 1))]
    for name, (used, sels, proj, bytes_per_row, keep) in cases.items():
        q = native.DeviceQuery(ctx, seg, used, sels, proj, 0, 1024)
        q.run()
        cnt = q.count()
        assert cnt == int(keep.sum()), (name, cnt, int(keep.sum()))
        q.reserve_rows(cnt + 1024)
        q.run()
        idx, _ = q.fetch_rows()
        assert (idx == np.flatnonzero(keep)).all(), name           # ordered rows (values are covered by tests/test_gpu_large.py)
        del idx
        dt = env.timed_steps(lambda i: q.run(), steps, 3) / steps
        k = env.kernel_times(q.run, steps)
        kms = {KERNEL_NAMES[i]: v for i, v in k.items()}
        ksum = sum(v for v in kms.values() if v)
        sel = cnt / n
        algo = bytes_per_row(sel) * n
        # The reference plans, runs and drops a pipeline per statement (Engine.scala:158-196): what ONE such statement costs here,
        # wall clock, handle creation and destruction included -- create (with its sampled selectivity estimate), one run, the
        # count (the first thing a consumer reads: it waits for the run), destroy.  Median of 7, after one unmeasured round.
        shots = []
        for r in range(8):
            env.sync()
            t0 = time.perf_counter()
            q1 = native.DeviceQuery(ctx, seg, used, sels, proj, 0, 1024)
            t1 = time.perf_counter()
            q1.run()
            c1 = q1.count()
            t2 = time.perf_counter()
            q1.close()
            t3 = time.perf_counter()
            assert c1 == cnt, (name, c1, cnt)
            if r:
                shots.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2))
        shots.sort()
        shot = shots[len(shots) // 2]
        out[name] = {
            "plan": q.plan(),
            "one_shot_ms": shot[0] * 1e3, "one_shot_parts_ms": {"create": shot[1] * 1e3, "run_and_count": shot[2] * 1e3, "destroy": shot[3] * 1e3},
            "rows_per_s": n / dt, "ms_per_query": dt * 1e3, "selected_rows": int(cnt), "selectivity": sel,
            "algorithmic_bytes_per_row": bytes_per_row(sel), "algorithmic_bytes": algo,
            "kernel_ms": kms, "kernel_ms_sum": ksum,
            "achieved_GBps": algo / (ksum * 1e-3) / 1e9,
            "frac": algo / (ksum * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_wall": algo / dt / 1e9 / HBM_PEAK_GBS,
        }
        # HBM bytes really moved (counters of a committed profiling session on these sources) next to the algorithmic ones: C4's
        # gathers fetch a 128-byte line per value, so its traffic-based fraction says how close its kernels are to what the memory
        # system can do, the algorithmic one what the query is worth
        tr = stamped_traffic(name)
        tb = tr.get("hbm_bytes_per_query")
        out[name]["roofline"] = {"bound": "hbm", "achieved": algo / (ksum * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": algo / (ksum * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tb, "traffic_source": tr.get("source"),
                                 "traffic_ratio": tb / algo if tb else None,
                                 "frac_traffic": tb / (ksum * 1e-3) / 1e9 / HBM_PEAK_GBS if tb else None}
        if with_cpu_baseline:
            out[name]["cpu_baseline"] = cpu_baseline_project([host_cols[u] for u in used], sels, proj, n, cnt)
        q.close()
    del keep3, keep4
    if not env.args.no_readme_table:
        out["readme_table_c3"] = readme_table_c3(env, ids, age, steps)
    # group-by aggregation: select count(id), max(age) from t [where age > 18 and age < 30] group by state
    for name, sels in (("agg_group_by_state_all_rows", []),
                       ("agg_group_by_state_range_age", [(1, native.GT, 18.0), (1, native.LT, 30.0)])):
        q = native.DeviceQuery(ctx, seg, [1, 2, 0], sels, (), 0, 1024, group_cols=[0], aggs=[(native.AGG_COUNT, 2), (native.AGG_MAX, 1)])
        dt = env.timed_steps(lambda i: q.run(), 10, 2) / 10
        k = env.kernel_times(q.run, 10, ids=(0, 3, 4))
        keys, first, counts, vals = q.fetch_groups()
        sel = float(counts.sum()) / n
        algo = ((1 if sels else 0) + 0.125 + sel * (2 + 1)) * n   # predicate column (age) + bitmap + sigma x (group key state + aggregated age)
        out[name] = {"rows_per_s": n / dt, "ms_per_query": dt * 1e3, "groups": int(keys.shape[0]), "selected_rows": int(counts.sum()),
                     "kernel_ms": {KERNEL_NAMES[i]: v for i, v in k.items()}, "algorithmic_bytes": algo,
                     "frac": algo / (sum(v for v in k.values() if v) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        q.close()
    # `limit` stops the scan (Project.scala:73-80): select id from t where id > T limit 10 -- met in the first megarow (T = 5), and in
    # the second half of the sorted key (T = 5e7: 200 MB of id lie before the first survivor, for the reference as for this path)
    for name, thr in (() if env.args.no_limit else (("limit10_met_in_the_first_rows", 5.0), ("limit10_met_in_the_second_half", float(n // 2)))):
        q = native.DeviceQuery(ctx, seg, [0], [(0, native.GT, thr)], [0], 10, 1024)
        dt = env.timed_steps(lambda i: q.run(), 10, 3) / 10
        k = env.kernel_times(q.run, 10, ids=(0, 1, 2), launches_per_run=16)
        idx, _ = q.fetch_rows()
        assert (idx == np.arange(int(thr) + 1, int(thr) + 11)).all()
        out[name] = {"query": f"select id from t where id > {int(thr)} limit 10", "ms_per_query": dt * 1e3,
                     "kernel_ms": {KERNEL_NAMES[i]: v for i, v in k.items()}, "kernel_ms_sum": sum(v for v in k.values() if v),
                     "note": "HIP events read ~3 us high per launch on launches that leave at once; profiles/ holds the rocprofv3 dispatch durations"}
        q.close()
    seg.close()
    # host -> HBM staging of 400 MB columns (the step before the path): synchronous create, and three asynchronous creates on
    # the context's copy stream while the query stream keeps scanning (what SegmentManager's start-up looks like here)
    off4 = synth.block_offsets(n, 4)
    t0 = time.perf_counter()
    s2 = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, ids.view(np.uint8), n * 4, off4)])
    dt = time.perf_counter() - t0
    q = native.DeviceQuery(ctx, s2, [0], [(0, native.GT, lo), (0, native.LT, hi)])
    scan_s = env.timed_steps(lambda i: q.run_select(), 20, 3) / 20
    hosts = [ids, ids + 1, ids + 2]                     # three distinct host buffers, as SegmentManager's mmaps are
    t0 = time.perf_counter()
    more = [native.DeviceSegment(ctx, [(native.DENSE_INT, 4, h.view(np.uint8), n * 4, off4)], async_copy=True) for h in hosts]
    t_enq = time.perf_counter() - t0                    # the three creates have returned: copies enqueued on the copy stream
    scans = 0
    while scans < 200 and time.perf_counter() - t0 < 3 * dt:   # meanwhile the query stream scans (bounded backlog)
        q.run_select()
        scans += 1
    for m in more:
        m.wait()
    t_copied = time.perf_counter() - t0                 # all three segments resident
    env.sync()
    out["staging_400MB"] = {"seconds": dt, "GBps": n * 4 / dt / 1e9, "note": "imm3_segment_create from pageable host memory, copy stream; PCIe link rate on this "
                            "platform is 55-56 GB/s for pageable, pinned and registered memory alike (tools/h2d_probe.py)",
                            "async_3_segments": {"create_calls_returned_after_s": t_enq, "resident_after_s": t_copied, "GBps": 3 * n * 4 / t_copied / 1e9,
                                                 "scans_enqueued_on_the_query_stream_meanwhile": scans, "scan_ms_alone": scan_s * 1e3,
                                                 "note": "imm3_segment_create_async: host ranges pinned in place, copies on the copy stream, scans keep running on the query stream"}}
    q.close()
    for m in more:
        m.close()
    s2.close()
    # PFOR_INT: the id column as PFORCodecInt.encode writes it; the range predicate is evaluated on the
    # compressed blocks (k_filter_pfor), HBM traffic = compressed bytes.  VALU-bound, not HBM-bound.
    dat, offs = native.pfor_encode_column(ids, 1024)
    sp = native.DeviceSegment(ctx, [(native.PFOR_INT, 4, dat, dat.size, offs)])
    q = native.DeviceQuery(ctx, sp, [0], [(0, native.GT, lo), (0, native.LT, hi)])
    q.run_select()
    cnt = q.count()
    dt = env.timed_steps(lambda i: q.run_select(), steps, 3) / steps
    k = env.kernel_times(q.run_select, steps, ids=(0, 3))
    out["pfor_range_id"] = {"rows_per_s": n / dt, "ms_per_query": dt * 1e3, "selected_rows": int(cnt), "compressed_bytes": int(dat.size),
                            "compression_ratio": n * 4 / dat.size, "kernel_ms": {KERNEL_NAMES[i]: v for i, v in k.items()},
                            "hbm_GBps": (dat.size + n / 8) / (k[0] * 1e-3) / 1e9 if k[0] else None, "bound": "valu"}
    q.close()
    sp.close()
    return out


# =====================================================================================================================
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=ROWS_PER_SEGMENT, help="rows per segment (default 100M)")
    ap.add_argument("--segments", type=int, default=3, help="distinct resident C2 segments rotated per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="N = 1: skip the extra block (C3, C4, aggregation, C5 at G = 1)")
    ap.add_argument("--no-graph", action="store_true", help="C5: launch every pass kernel by kernel instead of replaying the recorded hipGraph")
    ap.add_argument("--no-g1", action="store_true", help="N > 1: skip the solo leg (every rank runs the whole C5 job on its own GPU: the G = 1 point)")
    ap.add_argument("--no-c5", action="store_true", help="leave the C5 leg out of the extra block (profiling runs: its kernels are C3's instances)")
    ap.add_argument("--c5-per-segment", action="store_true", help="C5: one query (one launch) per owned segment, as round 4 ran it, instead of one table query per rank")
    ap.add_argument("--no-readme-table", action="store_true", help="leave the README-shaped 98-segment table out of the extra block")
    ap.add_argument("--no-limit", action="store_true", help="leave the `limit` queries out of the extra block (profiling runs: their chunks are launches of the headline kernel's instance)")
    ap.add_argument("--extra", action="store_true", help="(kept for compatibility: the extra block is on by default)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--grid", type=int, default=0)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus))            # before anything touches the GPU; the parent only relays rank 0's line
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: refusing to report a line for a different GPU count")

    env = Env(args)
    base = {"metric": METRIC, "unit": "rows/s", "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "vs_baseline": None, "data": "synthetic"}
    result = None
    # `value` is ONE workload at every N: the headline C2 step, every rank on its own segments (weak scaling)
    c2 = measure_c2(env, args.steps, args.warmup, with_cpu_baseline=(env.world == 1 and not args.no_cpu_baseline))
    if env.rank == 0:
        result = dict(base, scaling="weak", dtype="i32", **{k: c2[k] for k in ("value", "ms_per_step", "ms_per_step_with_kernel_events", "config", "roofline", "staging", "per_rank")})
        result["cpu_baseline"] = c2.get("cpu_baseline") if env.world == 1 else None     # timed at N = 1 only (contract)
    c5_keys = ("value", "ms_per_step", "launch", "ms_per_step_kernel_by_kernel", "ms_per_step_graph", "allreduce_gap_ms_per_pass", "allreduce_gap_note", "global_selected_rows_per_pass", "count_allreduce",
               "config", "roofline", "abandoned_runs", "busy_runs", "per_rank")
    if env.world == 1:
        if not args.no_extra:
            extra = extra_workloads(env, with_cpu_baseline=not args.no_cpu_baseline)
            if not args.no_c5:
                c5 = measure_c5(env, max(3, min(args.steps, 10)), 2, use_graph=not args.no_graph, with_cpu_baseline=not args.no_cpu_baseline)
                extra["c5"] = {k: c5[k] for k in c5_keys}
                extra["c5"]["cpu_baseline"] = c5.get("cpu_baseline")
                extra["c5"]["scaling"] = "strong"
                extra["c5"]["note"] = "BASELINE config C5 at G = 1: the first point of the strong-scaling curve whose G > 1 points are extra.c5.value of the --gpus N lines"
            result["extra"] = extra
    elif not args.no_c5:
        c5 = measure_c5(env, max(3, min(args.steps, 50)), min(args.warmup, 5), use_graph=not args.no_graph)
        g1 = None if args.no_g1 else measure_c5(env, max(3, min(args.steps, 10)), 2, use_graph=not args.no_graph, solo=True)
        if env.rank == 0:
            x = {k: c5[k] for k in c5_keys}
            x["scaling"] = "strong"
            if g1 is not None:
                x["g1_same_run"] = {"value": g1["value"], "ms_per_step": g1["ms_per_step"],
                                    "note": "the G = 1 point of this strong-scaling curve measured in this run: every rank ran the whole C5 job "
                                            "(8 segments) alone on its own GPU, time = max over ranks; no count collective"}
                x["efficiency"] = c5["value"] / (env.world * g1["value"])
            result["extra"] = {"c5": x}
    if env.rank == 0:
        result["scaling_note"] = ("value = the headline C2 workload at every N (weak scaling: every rank scans its own 100M-row segments; aggregate rows/s). "
                                  "BASELINE config C5 (8 segments sharded s mod N, strong scaling, RCCL count all-reduce) is extra.c5: its G = 1 point is "
                                  "extra.c5.value of the --gpus 1 line and extra.c5.g1_same_run.value of an N > 1 line; extra.c5.efficiency = value / (N x g1_same_run.value)")
    env.close()
    if env.rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
