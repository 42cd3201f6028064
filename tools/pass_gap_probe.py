#!/usr/bin/env python3
"""Development tool (round 5): what one count all-reduce per pass costs a rank whose pass is ONE launch over one 100 M-row segment (C5's
shape at G = 8): us per pass, wall, for (a) the launch alone, back to back, (b) launch + imm3_comm_allreduce_count (one-rank RCCL, not
waited for), and how long the host needs to enqueue a pass.  usage: pass_gap_probe.py [passes]"""
import sys
import time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = 100_000_000
ts = torch.cuda.Stream()
ctx = native.Context(0, ts.cuda_stream)
ids = np.arange(n, dtype=np.int32)
age = synth.uniform_below(2, n, 100, np.int8)
seg = native.DeviceSegment(ctx, [(native.DENSE_TINYINT, 1, age.view(np.uint8), n, synth.block_offsets(n, 1)),
                                 (native.DENSE_INT, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4))])
table = native.DeviceTable(ctx, [seg])
comm = native.Comm(ctx, 1, 0, native.comm_unique_id())
log = torch.zeros(K + 8, dtype=torch.int64, device="cuda")
sels = [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, 1e6), (1, native.LT, 9e7)]
want = int(((age > 18) & (age < 30) & (ids > 1e6) & (ids < 9e7)).sum())


def loop_logged(q):
    """(c) the run's own kernel logs its count (imm3_query_log_counts), the collective reads that word in place (imm3_comm_allreduce_u64):
    nothing but the launch on the context's stream"""
    big = torch.zeros(K + 16, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    q.log_counts(big.data_ptr(), K + 16)
    for i in range(3):
        q.run()
        comm.allreduce_u64(big.data_ptr() + 8 * i, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        q.run()
        comm.allreduce_u64(big.data_ptr() + 8 * (3 + i), 1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    q.log_counts(0, 0)
    assert big[:K + 3].tolist() == [want] * (K + 3)
    return (t2 - t0) / K * 1e6, (t1 - t0) / K * 1e6


def loop(q, with_comm):
    for i in range(3):
        q.run()
        if with_comm:
            comm.allreduce_count([q], device_out=log.data_ptr(), wait=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        q.run()
        if with_comm:
            comm.allreduce_count([q], device_out=log.data_ptr() + 8 * i, wait=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if with_comm:
        assert log[:K].tolist() == [want] * K
    return (t2 - t0) / K * 1e6, (t1 - t0) / K * 1e6


for name, src in (("segment query", seg), ("table of one segment", table)):
    q = native.DeviceQuery(ctx, src, [0, 1], sels, [1, 0], 0, 1024)
    q.run(); assert q.count() == want
    q.reserve_rows(want + 1024)
    q.run(); ctx.sync()
    ctx.timing_enable(64); ctx.timing_mask(1); ctx.timing_reset()
    for _ in range(10):
        q.run()
    ctx.sync()
    k_us = float(np.median(ctx.timing_collect(0))) * 1e3
    ctx.timing_enable(0)
    a, ha = loop(q, False)
    b, hb = loop(q, True)
    c, hc = loop_logged(q)
    print(f"{name:22s} kernel {k_us:6.1f} us | launch alone {a:6.1f} us per pass (host enqueue {ha:5.1f}) | + count all-reduce {b:6.1f} us per pass (host enqueue {hb:5.1f})"
          f" | + all-reduce of the logged count {c:6.1f} us per pass (host enqueue {hc:5.1f})", flush=True)
    q.close()
