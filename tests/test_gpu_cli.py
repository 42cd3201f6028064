"""GPU suite: the C++ host layer end to end -- imm3_sql (SQL text -> parser -> planner -> fused GPU pipelines per
segment -> `Row(...)` lines) against the committed golden expectations."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "immutable3_amd", "bin", "imm3_sql")
GOLDEN = os.path.join(ROOT, "tests", "golden")
EXPECTED = json.load(open(os.path.join(GOLDEN, "expected.json")))

SQL = {
    "c1_range_limit10": "select id, age from test_100 where (age > 18 and age < 30) limit 10",
    "c1_range_all": "select id, age from test_100 where (age > 18 and age < 30)",
    "c1_match_ca": "select id, state from test_100 where state = 'CA'",
    "c1_range_and_ca": "select id, state, age from test_100 where (age > 18 and age < 30 and state = 'CA')",
    "c1_eq_id": "select age, id from test_100 where id = 42",
    "c1_no_select": "select id from test_100 limit 5",
    "c1_tinyint_wrap": "select age from test_100 where age > 200 limit 3",
    "q25_range": "select id, age from quirk_25 where (id > 3 and id < 20)",
    "q25_match": "select id from quirk_25 where state = 'NY'",
    "q25_empty_batches": "select id from quirk_25 where id > 21",
}


def run(sql):
    p = subprocess.run([BIN, "-q", sql, "-d", GOLDEN], capture_output=True, text=True)
    return p.returncode, p.stdout.splitlines()


@pytest.mark.parametrize("name", sorted(SQL))
def test_sql_cli_matches_golden(name):
    e = EXPECTED[name]
    rows = [r for seg in e["segments"] for r in seg["rows"]]
    if e["limit"] > 0:
        rows = rows[: e["limit"]]
    rc, out = run(SQL[name])
    assert rc == 0
    assert out == ["Row(" + ",".join(str(v) for v in r) + ")" for r in rows]


def test_sql_cli_errors():
    rc, out = run("select id from test_100 where state > 3")
    assert rc == 1 and out == ["Unsupported column vector"]
    rc, out = run("select id from test_100 where age = 'CA'")
    assert rc == 1 and out == ["Unsupported column vector"]
    rc, out = run("select nope from test_100")
    assert rc == 1 and "does not exist" in out[0]
    rc, out = run("select id from missing")
    assert rc == 1 and "does not exist in SegmentManager" in out[0]


def test_sql_cli_aggregates():
    """select ... group by through the C++ Engine: per-segment GPU aggregation + combine by key."""
    rc, out = run("select count(id), max(age) from test_100 group by state")
    assert rc == 0
    import numpy as np
    from immutable3_amd import synth
    t = synth.test_100()
    age = t["age"].astype(int)
    assert out == [f"Row({len(range(k, 100, 7))},{age[k::7].max()}.0)" for k in range(7)]
    rc, out = run("select max(id), min(id), count(age) from quirk_25 where id > 2")
    assert rc == 0 and out == ["Row(24.0,3.0,22)"]
    rc, out = run("select min(state) from quirk_25")          # Min over STRING -> MaxStringAggr (Engine.scala:145)
    assert rc == 0 and out == ["Row(WA)"]
    rc, out = run("select sum(id) from quirk_25")
    assert rc == 1 and out == ["Unknown Aggregate type"]
