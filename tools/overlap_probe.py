#!/usr/bin/env python3
"""Development tool (round 5): what does a co-resident kernel on the communicator's stream do to a pass?

At G > 1 every pass ends with ONE ncclAllReduce of the count on the communicator's own stream, and the next pass's scans are
enqueued behind the count, not behind the collective (csrc/imm3_comm.cpp).  RCCL's kernel for it (ncclDevKernel_Generic_*, gfx950,
ROCm 7.2: 512 threads x 256 vector registers, 37 664 bytes of LDS) cannot share a CU with a work-group of the one-launch projection,
which wants all 256 CUs for the whole pass.  With one GPU there is no such kernel (a one-rank all-reduce launches none), so the
tools' build puts a stand-in of that footprint in front of every count all-reduce (imm3_comm_debug_standin) and this script
measures passes per second with it: (i) C3's one-launch query, (ii) the headline select (k_filter_tile, 512 work-groups), with and
without the CUs the library leaves free while a communicator is attached (tuning 16 switches the reservation off).

usage: IMM3_LIB_PATH=immutable3_amd/lib/libimm3_ablate.so python tools/overlap_probe.py [passes]"""
import sys
import time
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = 100_000_000
ts = torch.cuda.Stream()
ctx = native.Context(0, ts.cuda_stream)
ids = np.arange(n, dtype=np.int32)
age = synth.uniform_below(2, n, 100, np.int8)
seg = native.DeviceSegment(ctx, [(native.DENSE_TINYINT, 1, age.view(np.uint8), n, synth.block_offsets(n, 1)),
                                 (native.DENSE_INT, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4))])
vals = synth.uniform_int30(1, n)
seg2 = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, vals.view(np.uint8), n * 4, synth.block_offsets(n, 4))])
comm = native.Comm(ctx, 1, 0, native.comm_unique_id())
log = torch.zeros(K + 8, dtype=torch.int64, device="cuda")
sels3 = [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, 1e6), (1, native.LT, 9e7)]
want3 = int(((age > 18) & (age < 30) & (ids > 1e6) & (ids < 9e7)).sum())
want2 = int(((vals > 2 ** 28) & (vals < 3 * 2 ** 28)).sum())


def passes(q, project, want):
    """us per pass: K passes of run + count all-reduce (not waited for), wall clock; counts and flags checked afterwards."""
    for i in range(3):
        q.run() if project else q.run_select()
        comm.allreduce_count([q], device_out=log.data_ptr(), wait=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        q.run() if project else q.run_select()
        comm.allreduce_count([q], device_out=log.data_ptr() + 8 * i, wait=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e6
    assert log[:K].tolist() == [want] * K
    return dt


print(f"{K} passes each; us per pass (wall).  stand-in = work-groups x spin of a kernel with RCCL's footprint on the communicator's stream per pass")
print(f"{'workload':34s} {'reserved CUs':>12s} {'no stand-in':>12s} " + " ".join(f"{w}x{u}us".rjust(9) for w in (1, 2, 4) for u in (10, 30, 60)) + "   abandoned busy")
for name, project in (("C3 one launch (k_filter_project)", True), ("C2 select (k_filter_tile)", False)):
    for variant, label in ((16, "0"), (0, "1 per XCD")):
        if not project and variant == 16:
            continue
        ctx.set_tuning(variant, 0)
        comm.debug_standin(1, 0)
        if project:
            q = native.DeviceQuery(ctx, seg, [0, 1], sels3, [1, 0], 0, 1024)
            q.run(); assert q.count() == want3
            q.reserve_rows(want3 + 1024)
        else:
            q = native.DeviceQuery(ctx, seg2, [0], [(0, native.GT, float(2 ** 28)), (0, native.LT, float(3 * 2 ** 28))], [], 0, 1024)
            q.run_select(); assert q.count() == want2
        cells = []
        comm.debug_standin(1, 0)   # (a stand-in that leaves at once: the communicator counts as one whose collectives launch kernels, so the reservation applies)
        base = passes(q, project, want3 if project else want2)
        for w in (1, 2, 4):
            for u in (10, 30, 60):
                comm.debug_standin(w, u)
                cells.append(passes(q, project, want3 if project else want2))
        comm.debug_standin(0, 0)
        if project:
            idx, _ = q.fetch_rows()
            assert idx.size == want3
        p = q.plan()
        print(f"{name:34s} {label if project else '-':>12s} {base:12.1f} " + " ".join(f"{c:9.1f}" for c in cells) + f"   {p['abandoned_runs']:9d} {p['busy_runs']:4d}  grid {p['grid']}", flush=True)
        q.close()
        ctx.set_tuning(0, 0)
comm.close(); seg.close(); seg2.close(); ctx.close()
