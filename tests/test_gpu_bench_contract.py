"""GPU suite: bench.py prints ONE JSON line that carries the driver's contract fields, the roofline object and (when
asked) the cpu_baseline object.  Small sizes: this checks the shape of the line, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = run_bench("--rows", "2000000", "--steps", "10", "--warmup", "3", "--segments", "2", "--no-cpu-baseline", "--no-extra")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["unit"] == "rows/s" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["workload"].startswith("C2")
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0
    assert abs(d["value"] - 2000000 * 10 / (d["ms_per_step"] * 1e-3 * 10)) / d["value"] < 1e-6
    assert d["cpu_baseline"] is None   # --no-cpu-baseline


def test_bench_cpu_baseline_and_extra_block():
    """Default line: cpu_baseline plus the extra block -- C3, C4 (algorithmic bytes, per-kernel ms, frac), aggregation and
    the G = 1 point of the C5 curve (real one-rank RCCL count all-reduce inside libimm3)."""
    d = run_bench("--rows", "1000000", "--steps", "5", "--warmup", "2", "--segments", "2")
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "rows/s"
    x = d["extra"]
    for name in ("c3_range_age_id_project", "c4_match_state_project"):
        e = x[name]
        for k in ("algorithmic_bytes", "kernel_ms", "frac", "selected_rows", "ms_per_query"):
            assert k in e, (name, k)
        assert e["kernel_ms"]["scan_select"] > 0 and 0 < e["frac"] < 1.2
        # round 5: the reference's CPU operators timed beside every config, and counter traffic next to the algorithmic bytes
        cb = e["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["selected_rows"] == e["selected_rows"] and cb["nproc"] >= 1
        for k in ("traffic", "traffic_ratio", "frac_traffic", "traffic_source", "frac", "achieved", "peak"):
            assert k in e["roofline"], (name, k)
        if e["plan"]["single_pass"]:     # ONE launch: the filter kernel writes the rows (imm3_project.hip)
            assert e["plan"]["ran_single_pass"] and e["kernel_ms"]["compact_gather"] is None and e["kernel_ms"]["offsets_scan"] is None
        else:
            assert e["kernel_ms"]["compact_gather"] > 0
    # (at 1 M rows the cost model plans C3 as three small launches; at BASELINE's 100 M rows as ONE: tests/test_gpu_large.py)
    assert x["agg_group_by_state_all_rows"]["groups"] == 51
    c5 = x["c5"]
    assert c5["config"]["segments"] == 8 and c5["value"] > 0 and "ncclAllReduce" in c5["count_allreduce"]["collective"] and c5["scaling"] == "strong"
    assert c5["config"]["queries_per_gpu"] == 1                     # the rank's eight segments are ONE table query
    assert c5["cpu_baseline"]["cores"] >= 1 and c5["cpu_baseline"]["selected_rows"] == c5["global_selected_rows_per_pass"]
    assert "allreduce_gap_ms_per_pass" in c5 and "traffic" in c5["roofline"] and "frac_traffic" in c5["roofline"]
    rt = x["readme_table_c3"]
    assert rt["selected_rows"] == x["c3_range_age_id_project"]["selected_rows"] and rt["kernel_ms"]["scan_select"] > 0


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no rank environment starts two ranks itself and prints an n_gpus = 2 line whose `value`
    is the headline C2 workload (weak scaling) and whose extra.c5 is the sharded C5 job.  One-device rehearsal: both ranks share cuda:0, so the count all-reduce goes over gloo (RCCL refuses that)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["IMM3_BENCH_ONE_DEVICE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "1000000", "--steps", "4", "--warmup", "1",
                        "--segments", "2"], capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    # `value` is the headline workload at every N (weak scaling): like-for-like with the --gpus 1 line
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 4 and d["cpu_baseline"] is None
    assert d["config"]["workload"].startswith("C2") and d["dtype"] == "i32"
    assert abs(d["value"] - 2 * 1000000 * 4 / (d["ms_per_step"] * 1e-3 * 4)) / d["value"] < 1e-6
    assert [r["rank"] for r in d["per_rank"]] == [0, 1]
    # BASELINE config C5 (strong scaling, count all-reduce) rides in extra.c5, with its G = 1 point and an explicit efficiency
    c5 = d["extra"]["c5"]
    assert c5["scaling"] == "strong" and c5["config"]["segments"] == 8 and c5["config"]["segments_per_gpu"] == 4 and c5["config"]["workload"].startswith("C5")
    assert [r["segments"] for r in c5["per_rank"]] == [[0, 2, 4, 6], [1, 3, 5, 7]]
    assert c5["global_selected_rows_per_pass"] == sum(r["selected_rows"] for r in c5["per_rank"])
    g1 = c5["g1_same_run"]                       # the G = 1 point measured in the same run (every rank alone on all 8 segments)
    assert g1["value"] > 0 and abs(g1["value"] - 8 * 1000000 / (g1["ms_per_step"] * 1e-3)) / g1["value"] < 1e-6
    assert abs(c5["efficiency"] - c5["value"] / (2 * g1["value"])) < 1e-9
    # a rank count that contradicts the environment is refused, not silently reported as another N
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "100000"], capture_output=True, text=True, cwd=ROOT, timeout=300, env=env2)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_count_log():
    """imm3_query_log_counts: every run's count lands in the device log without a host call."""
    import numpy as np
    import torch
    from immutable3_amd import native, synth
    with pytest.raises(ValueError):
        native.Context(0, torch.cuda.current_stream().cuda_stream)     # 0 = torch's default stream: refused loudly
    ts = torch.cuda.Stream()
    ctx = native.Context(0, ts.cuda_stream)                            # the library's work rides a torch stream
    assert ctx.stream == ts.cuda_stream
    n = 300_000
    v = synth.uniform_int30(7, n)
    seg = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, v.view(np.uint8), n * 4, synth.block_offsets(n, 4))])
    q1 = native.DeviceQuery(ctx, seg, [0], [(0, native.GT, float(2 ** 28))])                               # single tile pass: count in the kernel
    q2 = native.DeviceQuery(ctx, seg, [0], [(0, native.GT, float(2 ** 28)), (0, native.LT, float(2 ** 29))], [0], 0)  # with projection
    for q, want in ((q1, int((v > 2 ** 28).sum())), (q2, int(((v > 2 ** 28) & (v < 2 ** 29)).sum()))):
        log = torch.zeros(6, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        q.log_counts(log.data_ptr(), 4)
        for _ in range(5):
            q.run_select()
        ts.synchronize()                                    # torch's handle on the same stream covers the library's launches
        assert log.tolist() == [want] * 4 + [0, 0]          # capacity 4: the fifth run is not logged
        q.log_counts(0, 0)
        q.run_select()
        assert q.count() == want
        q.close()
    seg.close()
    # a multi-pass chain (four predicate columns -> two tile passes) reduces its count in k_total: logged there
    cols = [synth.uniform_int30(20 + c, n) for c in range(4)]
    seg = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, c.view(np.uint8), n * 4, synth.block_offsets(n, 4)) for c in cols])
    q = native.DeviceQuery(ctx, seg, [0, 1, 2, 3], [(c, native.GT, float(2 ** 27)) for c in range(4)])
    want = int(np.logical_and.reduce([c > 2 ** 27 for c in cols]).sum())
    log = torch.zeros(3, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    q.log_counts(log.data_ptr(), 3)
    for _ in range(2):
        q.run_select()
    ts.synchronize()
    assert log.tolist() == [want, want, 0] and q.count() == want
    q.close()
    seg.close()
    ctx.close()
