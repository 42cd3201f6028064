"""GPU suite: the WORLD > 1 paths of the communicator (csrc/imm3_comm.cpp) with two ranks -- the count all-reduce
(Engine.scala:190-196: the per-segment pipelines' counts meet in one place) and the cross-rank group merge
(ProjectAggregateQueueOp, ProjectAggregateQueue.scala:9-55): the 5-word shape / error exchange, the list-length flag, the
allocation flag, ncclAllGather of the packed lists, the second hash table, and "every rank takes the same exit" when ONE rank
fails before or between the collectives.

RCCL refuses two ranks on one device and the development boxes have one GPU, so until round 5 every collective here had only
ever run with one rank.  The ranks of this test are two THREADS of one process, each with its own context on device 0, and the
nine nccl* entry points the library binds at run time come from tests/native/loopback_rccl.cpp (IMM3_RCCL_LIB): a transport
that meets at a barrier and reduces through host memory.  It is the library's protocol that is under test, not RCCL; the real
RCCL runs in the one-rank tests (test_gpu_lifetime.py, test_gpu_agg.py) and in the driver's multi-GPU bench."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "loopback_rccl.cpp")

WORKER = r'''
import sys, threading
import numpy as np
import torch  # noqa: F401  (its HIP runtime first: conftest.py says why)
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from conftest import DENSE_INT, DENSE_TINYINT, GT, LT, RawColumn, blocks_of
from immutable3_amd import native
from oracle import oracle_np

WORLD = 2
ctxs = [native.Context(0) for _ in range(WORLD)]
uid = native.comm_unique_id()
comms = [None] * WORLD

def both(fn):
    """Run fn(rank) on two threads (the two ranks); returns their results, re-raises what they raised."""
    out, err = [None] * WORLD, [None] * WORLD
    def run(r):
        try:
            out[r] = fn(r)
        except BaseException as e:      # noqa: BLE001
            err[r] = e
    ts = [threading.Thread(target=run, args=(r,)) for r in range(WORLD)]
    for t in ts: t.start()
    for t in ts: t.join(120)
    assert not any(t.is_alive() for t in ts), "a rank is stuck in a collective: the ranks did not take the same exit"
    return out, err

def mk(r):
    comms[r] = native.Comm(ctxs[r], WORLD, r, uid)
out, err = both(mk)
assert err == [None, None], err

# ---- data: four segments, segment s on rank s mod 2 (the partition of Engine.scala:176-180 over two GPUs)
rng = np.random.default_rng(11)
N_SEG = 4
rows = [70_000, 50_001, 1024, 33_333]
cols = []
for s, n in enumerate(rows):
    k8 = rng.integers(-20, 20, size=n).astype(np.int8)           # narrow key (direct table, element-wise all-reduces)
    k32 = rng.integers(0, 3000, size=n).astype(np.int32) * 7      # wide key (hash tables, all-gather of packed lists)
    val = rng.integers(-10 ** 6, 10 ** 6, size=n).astype(np.int32)
    br = blocks_of(n, 1024)
    cols.append([RawColumn(DENSE_TINYINT, 1, k8, br), RawColumn(DENSE_INT, 4, k32, br), RawColumn(DENSE_INT, 4, val, br)])
owner = [s % WORLD for s in range(N_SEG)]
segs = [native.DeviceSegment(ctxs[owner[s]], [c.native() for c in cols[s]]) for s in range(N_SEG)]
KIND = {"count": native.AGG_COUNT, "min": native.AGG_MIN, "max": native.AGG_MAX}

def expected(group, aggs, sels):
    per_seg = []
    for s in range(N_SEG):
        _, _, masks = oracle_np.scan_select([c.npcol() for c in cols[s]], sels, 1024)
        per_seg.append(oracle_np.project_agg([c.npcol() for c in cols[s]], group, aggs, masks))
    return oracle_np.combine_agg(per_seg, aggs)

def decode(keys, counts, vals, group, aggs):
    got = []
    for g in range(keys.shape[0]):
        raw = int(keys[g]).to_bytes(8, "little")
        parts, off = [], 0
        for gi in group:
            w = cols[0][gi].width
            parts.append(str(int.from_bytes(raw[off: off + w], "little", signed=True)))
            off += w
        got.append(("_".join(parts), [int(counts[g]) if k == "count" else float(int(vals[g, j])) for j, (k, _) in enumerate(aggs)]))
    return got

for group, aggs, sels in (([0], [("count", 2), ("max", 2), ("min", 2)], []),                     # narrow key: all-reduces of a direct table
                          ([1], [("count", 2), ("max", 2)], []),                                  # wide key: lists, all-gather, second table
                          ([0, 1], [("count", 0), ("min", 2)], [(2, GT, 0.0)]),                   # 5-byte key, with a predicate
                          ([1], [("count", 2)], [(2, GT, 2.0e6)])):                               # nothing survives anywhere: empty lists
    queries = [[], []]
    seg_idx = [[], []]
    for s in range(N_SEG):
        q = native.DeviceQuery(ctxs[owner[s]], segs[s], [0, 1, 2], sels, (), 0, 1024, group_cols=group, aggs=[(KIND[k], c) for k, c in aggs])
        q.run()
        queries[owner[s]].append(q)
        seg_idx[owner[s]].append(s)
    out, err = both(lambda r: comms[r].merge_groups(queries[r], seg_idx[r]))
    assert err == [None, None], err
    want = [(k, v) for k, v in expected(group, aggs, sels).items()]
    for r in range(WORLD):
        keys, first, counts, vals = out[r]
        assert decode(keys, counts, vals, group, aggs) == want, ("rank", r, group, aggs)
    assert out[0][1].tolist() == out[1][1].tolist()                 # every rank has the same table, in the same (first-seen) order
    # the count all-reduce over two ranks (its own stream, fenced by events)
    out, err = both(lambda r: comms[r].allreduce_count(queries[r]))
    assert err == [None, None], err
    want_count = sum(int(q.count()) for qs in queries for q in qs)
    assert out == [want_count, want_count], (out, want_count)
    print("merge ok", group, aggs, "groups", len(want), flush=True)

    # ---- one rank fails BEFORE the first collective (a query of another context): both ranks must come back, with an error each
    bad = [list(queries[0]), [queries[0][0]]]                       # rank 1 brings rank 0's query
    out, err = both(lambda r: comms[r].merge_groups(bad[r], seg_idx[r][: len(bad[r])]))
    assert isinstance(err[0], native.Imm3Error) and isinstance(err[1], native.Imm3Error), (out, err)
    assert "another rank failed" in str(err[0]) and "another context" in str(err[1]), err
    # ... and the communicator still works afterwards
    out, err = both(lambda r: comms[r].merge_groups(queries[r], seg_idx[r]))
    assert err == [None, None] and decode(out[1][0], out[1][2], out[1][3], group, aggs) == want, err
    for qs in queries:
        for q in qs:
            q.close()

# ---- the ranks disagree on the shape (different aggregates): every rank sees it in the same all-reduce and leaves
qa = native.DeviceQuery(ctxs[0], segs[0], [0, 1, 2], [], (), 0, 1024, group_cols=[1], aggs=[(native.AGG_COUNT, 2)])
qb = native.DeviceQuery(ctxs[1], segs[1], [0, 1, 2], [], (), 0, 1024, group_cols=[1], aggs=[(native.AGG_COUNT, 2), (native.AGG_MAX, 2)])
qa.run(); qb.run()
out, err = both(lambda r: comms[r].merge_groups([qa, qb][r: r + 1], [r]))
assert all(isinstance(e, native.Imm3Error) and "differ" in str(e) for e in err), err
qa.close(); qb.close()
print("failure exits ok", flush=True)
for c in comms: c.close()
for s in segs: s.close()
for c in ctxs: c.close()
print("LOOPBACK-OK", flush=True)
'''


def test_two_ranks_merge_and_count_over_the_loopback_transport(tmp_path):
    lib = tmp_path / "libloopback_rccl.so"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", SRC, "-o", str(lib)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    script = tmp_path / "loopback_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, IMM3_RCCL_LIB=str(lib))
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=900)
    sys.stdout.write(r.stdout[-4000:])
    sys.stderr.write(r.stderr[-4000:])
    assert r.returncode == 0 and "LOOPBACK-OK" in r.stdout
