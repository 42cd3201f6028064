#!/usr/bin/env python3
"""Development tool: builds a README-style table (block 1024 x segment 1000 -> ~1 M-row segments incl. the loader's
trailing 1-row block) under /tmp and times SQL queries through the C++ CLI (imm3_sql) end to end."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import synth
from immutable3_amd.schema import TableIO
from immutable3_amd.storage import write_segment_arrays

root = "/tmp/imm3_multiseg"
n_seg = int(sys.argv[1]) if len(sys.argv) > 1 else 98
t = synth.table_schema("t100m", 1024)
if not os.path.exists(os.path.join(root, "t100m", "_table.meta")):
    TableIO.store(root, t)
    rows = 1024 * 1000 + 1                      # a "full" loader segment: S*B + 1 rows, last block has 1 row
    for s in range(n_seg):
        n = rows
        cols = {"id": (np.arange(n, dtype=np.int64) + s * rows).astype(np.int32), "age": synth.uniform_below(100 + s, n, 100, np.int8),
                "state": synth.state_codes(200 + s, n)}
        write_segment_arrays(root, t, s, cols, block_rows=[1024] * 1000 + [1])
    print("table written")
binp = os.path.join(__file__.rsplit("/tools/", 1)[0], "immutable3_amd", "bin", "imm3_sql")
for sql in ("select count(id) from t100m where (age > 18 and age < 30)",
            "select id, age from t100m where (age > 18 and age < 30) limit 10",
            "select count(id), max(age) from t100m group by state"):
    t0 = time.perf_counter()
    p = subprocess.run([binp, "-q", sql, "-d", root, "--repeat", "5"], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    rep = [l.split(":")[1].split("ms")[0].strip() for l in p.stderr.splitlines() if l.startswith("repeat")]
    print(f"{dt*1e3:8.1f} ms process  rc={p.returncode}  resident repeats (ms): {rep}  {sql}  -> {p.stdout.splitlines()[:2]}")
