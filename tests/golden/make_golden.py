#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

The reference ships no fixtures and cannot run here (Scala/JVM, no JDK), so these vectors are NOT produced by
the reference: the table files are written by immutable3_amd.storage (the restated SegmentWriter / loader,
including its block-layout quirk) and the expected outputs by the two independent CPU restatements
(oracle/imm3_oracle.c and oracle/oracle_np.py), which must agree bit-for-bit before anything is written.
The test_100 expectations are additionally pinned to the hand-derived values of SURVEY.md Appendix B8.

    python tests/golden/make_golden.py        # rewrites tests/golden/{test_100,quirk_25,expected.json}
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from immutable3_amd import synth  # noqa: E402
from immutable3_amd.schema import CodecType  # noqa: E402
from immutable3_amd.storage import SegmentManager, load_rows  # noqa: E402
from oracle import oracle_c, oracle_np  # noqa: E402

GT, LT, EQ, MATCH = oracle_c.GT, oracle_c.LT, oracle_c.EQ, oracle_c.MATCH

# (name, table, used column names in Engine.getColumns order, select leaves (col, cond, operand), project cols, limit)
QUERIES = [
    ("c1_range_limit10", "test_100", ["age", "id"], [("age", GT, 18.0), ("age", LT, 30.0)], ["id", "age"], 10),
    ("c1_range_all", "test_100", ["age", "id"], [("age", GT, 18.0), ("age", LT, 30.0)], ["id", "age"], 0),
    ("c1_match_ca", "test_100", ["state", "id"], [("state", MATCH, ["CA"])], ["id", "state"], 0),
    ("c1_range_and_ca", "test_100", ["age", "state", "id"], [("age", GT, 18.0), ("age", LT, 30.0), ("state", MATCH, ["CA"])], ["id", "state", "age"], 0),
    ("c1_eq_id", "test_100", ["id", "age"], [("id", EQ, 42.0)], ["age", "id"], 0),
    ("c1_in_list", "test_100", ["state", "age"], [("state", MATCH, ["NY", "TX", "ZZ", "CAL"])], ["state", "age"], 7),
    ("c1_no_select", "test_100", ["id"], [], ["id"], 5),
    ("c1_tinyint_wrap", "test_100", ["age"], [("age", GT, 200.0)], ["age"], 3),      # GT(200) on TINYINT == > -56
    ("q25_range", "quirk_25", ["id", "age"], [("id", GT, 3.0), ("id", LT, 20.0)], ["id", "age"], 0),
    ("q25_match", "quirk_25", ["state", "id"], [("state", MATCH, ["NY"])], ["id"], 0),
    ("q25_empty_batches", "quirk_25", ["id"], [("id", GT, 21.0)], ["id"], 0),
]


def build_tables():
    for name in ("test_100", "quirk_25"):
        shutil.rmtree(os.path.join(HERE, name), ignore_errors=True)
    t = synth.test_100()
    rows = [[str(int(t["id"][i])), bytes(t["state"][i]).decode(), str(int(t["age"][i]))] for i in range(100)]
    load_rows(HERE, synth.table_schema("test_100", 1024), rows, segmentSize=100)          # 1 segment, blocks [100]
    # loader quirk: 25 rows, blockSize 4, segmentSize 2 -> segments [4,4,1] [4,4,1] [4,3]
    rows = [[str(i), synth.CODES7[i % 7], str((i * 5) % 11 - 5)] for i in range(25)]
    load_rows(HERE, synth.table_schema("quirk_25", 4), rows, segmentSize=2)


def run_query(sm, table, used, leaves, proj, limit):
    t = sm.getTable(table)
    out = []
    for seg in range(sm.getTableSegmentCount(table)):
        ocols, ncols = [], []
        for cname in used:
            c = t.getColumn(cname)
            dat = np.asarray(sm.segments[f"{table}.{cname}"][seg])
            offs = sm.segmentsMeta[f"{table}.{cname}"][seg].blockOffsets
            ocols.append(oracle_c.OColumn(dat, offs, CodecType.id_of(c.codec), c.width))
            ncols.append((dat, offs, CodecType.id_of(c.codec), c.width))
        sels = [(used.index(cn), cond, ([v.encode() for v in op] if cond == MATCH else op)) for (cn, cond, op) in leaves]
        pj = [used.index(cn) for cn in proj]
        wc, cc = oracle_c.scan_select(ocols, sels, t.blockSize, 0)
        wt, ct = oracle_c.scan_select(ocols, sels, t.blockSize, 1)
        wn, cn_, masks = oracle_np.scan_select(ncols, sels, t.blockSize)
        assert cc == ct == cn_ and wc.tolist() == wt.tolist() == wn.tolist(), "oracles disagree"
        n, batch, pos, vals, would_throw = oracle_c.project(ocols, pj, limit, t.blockSize, wc)
        rows_np, where, _ = oracle_np.project(ncols, pj, limit, masks)
        rows = []
        for i in range(n):
            r = []
            for j, p in enumerate(pj):
                c = ocols[p]
                if c.codec == oracle_c.DENSE_INT:
                    r.append(int(vals[j][i].view("<i4")[0]))
                elif c.codec == oracle_c.DENSE_TINYINT:
                    r.append(int(vals[j][i].view(np.int8)[0]))
                else:
                    r.append(bytes(vals[j][i]).decode())
            rows.append(r)
        assert [list(x) if not isinstance(x, list) else x for x in rows] == [
            [v.decode() if isinstance(v, bytes) else v for v in rr] for rr in rows_np], "oracles disagree on rows"
        size, oid, woff, _ = oracle_c.layout(ocols[0], t.blockSize)
        out.append({
            "segment": seg, "batch_size": size.tolist(), "batch_oid": oid.tolist(), "batch_word_off": woff.tolist(),
            "words_hex": [f"{int(w):016x}" for w in wc], "count": cc, "rows": rows,
            "row_batch": batch.tolist(), "row_pos": pos.tolist(), "reference_would_throw_on_empty_batch": bool(would_throw),
        })
    return out


def main():
    oracle_c.build()
    build_tables()
    sm = SegmentManager(HERE)
    expected = {}
    for name, table, used, leaves, proj, limit in QUERIES:
        expected[name] = {"table": table, "used": used, "select": [[c, cond, op if cond != MATCH else list(op)] for c, cond, op in leaves],
                          "project": proj, "limit": limit, "segments": run_query(sm, table, used, leaves, proj, limit)}
    # pin to SURVEY Appendix B8 (hand-derived): fail loudly if the restatements ever drift
    b8 = expected["c1_range_limit10"]["segments"][0]
    assert b8["words_hex"] == ["0042100108008400", "0000000001080084"] and b8["count"] == 11
    assert b8["rows"] == [[10, 21], [15, 26], [27, 20], [32, 25], [44, 19], [49, 24], [54, 29], [66, 23], [71, 28], [83, 22]]
    assert [r[0] for r in expected["c1_match_ca"]["segments"][0]["rows"]] == list(range(0, 100, 7))
    assert expected["c1_range_and_ca"]["segments"][0]["rows"] == [[49, "CA", 24]]
    with open(os.path.join(HERE, "expected.json"), "w") as f:
        json.dump(expected, f, indent=1)
    print("wrote", os.path.join(HERE, "expected.json"))


if __name__ == "__main__":
    main()
