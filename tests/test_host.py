"""CPU suite: host logic (format writer/reader, planner) and the C-ABI library's exported surface."""
import ctypes
import os
import re

import numpy as np
import pytest

from immutable3_amd import EQ, GT, LT, And, Match, NoSelect, Or, Project, Query, Select
from immutable3_amd import native, synth
from immutable3_amd.operators import getColumns, resolveSelectOps, SelectOp
from immutable3_amd.schema import CodecType, Column, Table, TableIO
from immutable3_amd.storage import SegmentManager, SegmentMeta, SegmentWriter, load_csv, load_rows, write_segment_arrays

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_header_symbol():
    L = native.load()
    for header, names in (("imm3.h", native.EXPORTS), ("imm3_diag.h", native.DIAG_EXPORTS)):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        declared = sorted(set(re.findall(r"\b(imm3_[a-z0-9_]+)\s*\(", hdr)))
        assert declared == sorted(names), (header, set(declared) ^ set(names))
        for name in declared:
            assert isinstance(getattr(L, name), ctypes._CFuncPtr)
    # the drop-in boundary carries no measurement / tuning hooks
    assert not set(native.EXPORTS) & set(native.DIAG_EXPORTS)
    assert L.imm3_abi_version() == 1
    assert isinstance(native.device_count(), int)


def test_jni_shim_syntax_checks_against_a_stub_header():
    """integration/jni/imm3_jni.c cannot be built here (no JDK); at least its C parses and type-checks against include/imm3.h
    with a minimal stand-in for jni.h (tests/jni_stub/jni.h).  Never linked, never run: the shim stays UNVERIFIED."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    p = subprocess.run([gcc, "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-DIMM3_HAVE_JNI=1",
                        "-I" + os.path.join(ROOT, "tests", "jni_stub"), "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "integration", "jni", "imm3_jni.c")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    # every @native method of Native.scala has its Java_immutabledb_gpu_Native_00024_<name> in the shim
    scala = open(os.path.join(ROOT, "integration", "scala", "immutabledb", "gpu", "Native.scala")).read()
    shim = open(os.path.join(ROOT, "integration", "jni", "imm3_jni.c")).read()
    natives = re.findall(r"@native def (\w+)", scala)
    assert natives and all(f"Java_immutabledb_gpu_Native_00024_{n}(" in shim for n in natives), [n for n in natives if f"_{n}(" not in shim]


def test_no_gpu_fails_loudly_not_silently():
    """Without a HIP device the product path raises; it never falls back to a CPU evaluation."""
    if native.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(native.Imm3Error) as e:
        native.Context(0)
    assert e.value.code == native.ERR_DEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "immutable3_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src and "imm3_oracle" not in src, fn


def test_table_meta_roundtrip(tmp_path):
    t = synth.table_schema("t1", 256)
    TableIO.store(str(tmp_path), t)
    t2 = TableIO.load(str(tmp_path), "t1")
    assert t2 == t and t2.getColumn("state").width == 2 and t2.getColumn("id").width == 4
    with pytest.raises(Exception, match="does not exist"):
        t2.getColumn("nope")


def test_loader_split_drops_trailing_empty_fields_like_java(tmp_path):
    """LoaderCli splits with java.lang.String.split (LoaderCli.scala:136): "1,CA," binds two fields, not three."""
    from immutable3_amd.storage import java_split_comma
    assert java_split_comma("1,CA,") == ["1", "CA"] and java_split_comma(",,") == [] and java_split_comma("") == [""]
    assert java_split_comma("1,,3") == ["1", "", "3"] and java_split_comma("1,CA, ") == ["1", "CA", " "]
    t = Table("tr", [Column.make("id", CodecType.DENSE_INT), Column.make("state", CodecType.DENSE_STRING, {"size": "2"}),
                     Column.make("age", CodecType.DENSE_TINYINT)], 4)
    csv = tmp_path / "in.csv"
    csv.write_text("id,state,age\n1,CA,5\n2,NY,\n3,TX,7,\n")     # row 2 has no age: the reference writes nothing for it
    load_csv(str(tmp_path), t, str(csv), 10)
    sm = SegmentManager(str(tmp_path))
    ids = np.asarray(sm.getSegment(0, "tr", "id").segmentData).view("<i4")
    ages = np.asarray(sm.getSegment(0, "tr", "age").segmentData).view(np.int8)
    assert ids.tolist() == [1, 2, 3] and ages.tolist() == [5, 7]        # the age column silently falls one row short, as in the reference


def test_loader_csv_and_lexicographic_segment_order(tmp_path):
    # 12 segments -> files _0 .. _11; SegmentManager sorts by NAME, so _10 and _11 come before _2
    t = Table("lex", [Column.make("id", CodecType.DENSE_INT)], 2)
    csv = tmp_path / "in.csv"
    csv.write_text("id\n" + "\n".join(f" {i} " for i in range(58)) + "\n")     # header skipped, fields trimmed
    load_csv(str(tmp_path), t, str(csv), segmentSize=2)                         # 5 rows per full segment (2*2+1)
    sm = SegmentManager(str(tmp_path))
    n = sm.getTableSegmentCount("lex")
    assert n == 12
    firsts = [int(np.asarray(d).view("<i4")[0]) for d in sm.segments["lex.id"]]
    names = sorted(f"id_{i}.dat" for i in range(12))
    assert firsts == [int(nm.split("_")[1].split(".")[0]) * 5 for nm in names]
    assert firsts[:4] == [0, 5, 50, 55]                                          # _0, _1, _10, _11


def test_segment_writer_errors(tmp_path):
    t = Table("w", [Column.make("a", CodecType.DENSE_TINYINT)], 4)
    w = SegmentWriter(0, 4, "w", t.columns[0], str(tmp_path), 2)
    with pytest.raises(ValueError):
        w.write("128")                                  # "128".toByte -> NumberFormatException
    w.write("-128")
    w.close()
    assert open(tmp_path / "w" / "a_0.dat", "rb").read() == b"\x80"


def test_bulk_writer_matches_loader(tmp_path):
    t = synth.table_schema("bulk", 16)
    a = synth.test_100()
    write_segment_arrays(str(tmp_path / "x"), t, 0, a)
    rows = [[str(int(a["id"][i])), bytes(a["state"][i]).decode(), str(int(a["age"][i]))] for i in range(100)]
    load_rows(str(tmp_path / "y"), t, rows, segmentSize=1000)
    TableIO.store(str(tmp_path / "x"), t)
    for c in ("id", "state", "age"):
        assert open(tmp_path / "x" / "bulk" / f"{c}_0.dat", "rb").read() == open(tmp_path / "y" / "bulk" / f"{c}_0.dat", "rb").read()
        assert open(tmp_path / "x" / "bulk" / f"{c}_0.meta").read() == open(tmp_path / "y" / "bulk" / f"{c}_0.meta").read()


def test_get_columns_order_and_select_flattening():
    t = synth.table_schema("t")
    q = Query("t", And(Select("age", GT(18)), Select("age", LT(30))), Project(["id", "age"], 10))
    assert [c.name for c in getColumns(q, t)] == ["age", "id"]
    q = Query("t", And(Select("age", GT(0)), Or(Select("state", Match(["VA"])), Select("id", EQ(3)))), Project(["id", "state"]))
    assert [c.name for c in getColumns(q, t)] == ["age", "state", "id"]
    leaves = resolveSelectOps(q)
    ops = [leaf(None) for leaf in leaves]
    assert [(o.col, type(o.cond).__name__) for o in ops] == [("age", "GT"), ("state", "Match"), ("id", "EQ")]
    q = Query("t", NoSelect, Project(["state"]))
    assert [c.name for c in getColumns(q, t)] == ["state"] and resolveSelectOps(q) == []


def test_splitmix64_reference_values():
    # first outputs of splitmix64 seeded with 1 (Vigna's reference implementation)
    assert int(synth.splitmix64(1, 1)[0]) == 0x910A2DEC89025CC1
    assert int(synth.splitmix64(0, 1)[0]) == 0xE220A8397B1DCDAF
    assert synth.splitmix64(7, 5, 3).tolist() == synth.splitmix64(7, 8)[3:].tolist()
    v = synth.uniform_int30(1, 1000)
    assert v.min() >= 0 and v.max() < 2 ** 30
    assert synth.block_offsets(100_000_000, 4, 1024).size == 97657 + 1 and int(synth.block_offsets(2500, 4)[-1]) == 10000


def test_pfor_segment_writer_matches_oracle_encoder(tmp_path):
    """A PFOR_INT column goes through codec.encode per block (Segment.scala:115-122 -> PFORCodec.scala:19-31): the
    row-at-a-time SegmentWriter, the bulk writer and the C++ loader CLI all write the oracle encoder's bytes."""
    import subprocess
    from oracle import oracle_c
    from immutable3_amd.build import build_native
    build_native()
    vals = [(i * 37) % 1000 - 300 if i % 50 else -7 for i in range(150)]
    t = Table("pf", [Column.make("v", CodecType.PFOR_INT)], 64)
    load_rows(str(tmp_path / "a"), t, [[str(v)] for v in vals], segmentSize=100)
    write_segment_arrays(str(tmp_path / "b"), t, 0, {"v": np.array(vals, dtype=np.int32)})
    csv = tmp_path / "in.csv"
    csv.write_text("v\n" + "\n".join(str(v) for v in vals) + "\n")
    subprocess.check_call([os.path.join(ROOT, "immutable3_amd", "bin", "imm3_loader"), "-t", "pf", "-c", "v:PFOR_INT", "-d", str(tmp_path / "c"),
                           "-i", str(csv), "--block-size", "64", "--segment-size", "100"])
    want = b"".join(oracle_c.pfor_encode_block(np.array(vals[s:s + 64], dtype=np.int32)) for s in range(0, 150, 64))
    for d in "abc":
        assert open(tmp_path / d / "pf" / "v_0.dat", "rb").read() == want, d
    offs = SegmentMeta.load(str(tmp_path / "a" / "pf" / "v_0.meta")).blockOffsets
    assert len(offs) == 4 and offs[-1] == len(want)
    assert open(tmp_path / "a" / "pf" / "v_0.meta").read() == open(tmp_path / "c" / "pf" / "v_0.meta").read()


def test_snappy_writer_is_readable_by_the_oracle_and_google_snappy(tmp_path):
    """The product's SnappyCodec.encode restatement (host/codec.hpp): its blocks decode with the oracle's reader, and the
    raw payload of a compressed chunk decodes with pyarrow's Google snappy.  Writers (Python row-at-a-time, bulk, C++
    loader) agree byte for byte."""
    import subprocess
    from oracle import oracle_c
    from immutable3_amd.build import build_native
    build_native()
    rng = np.random.default_rng(4)
    for raw in (b"", b"a", bytes(5000), rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(), (np.arange(20000) // 9).astype("<i4").tobytes(),
                b"CANYTX" * 7000):
        blk = native.snappy_encode_block(raw)
        assert blk[:7] == b"snappy\x00" and oracle_c.snappy_block_decode(blk) == raw
        if raw and blk[7] == 1:
            pa = pytest.importorskip("pyarrow")
            plen = int.from_bytes(blk[8:10], "big")
            first = raw[:32768]
            assert pa.Codec("snappy").decompress(blk[14:14 + plen], decompressed_size=len(first), asbytes=True) == first
    vals = [str((i * 37) % 1000 // 10) for i in range(300)]
    t = Table("sn", [Column.make("v", CodecType.SNAPPY_INT), Column.make("s", CodecType.SNAPPY_STRING, {"size": "2"})], 128)
    rows = [[v, synth.CODES7[i % 7]] for i, v in enumerate(vals)]
    load_rows(str(tmp_path / "a"), t, rows, segmentSize=100)
    write_segment_arrays(str(tmp_path / "b"), t, 0, {"v": np.array([int(v) for v in vals], dtype=np.int32),
                                                     "s": np.array([list(synth.CODES7[i % 7].encode()) for i in range(300)], dtype=np.uint8)})
    csv = tmp_path / "in.csv"
    csv.write_text("v,s\n" + "\n".join(",".join(r) for r in rows) + "\n")
    subprocess.check_call([os.path.join(ROOT, "immutable3_amd", "bin", "imm3_loader"), "-t", "sn", "-c", "v:SNAPPY_INT,s:SNAPPY_STRING:size=2",
                           "-d", str(tmp_path / "c"), "-i", str(csv), "--block-size", "128", "--segment-size", "100"])
    for f in ("v_0.dat", "v_0.meta", "s_0.dat", "s_0.meta"):
        ref = open(tmp_path / "a" / "sn" / f, "rb").read()
        assert ref == open(tmp_path / "b" / "sn" / f, "rb").read() == open(tmp_path / "c" / "sn" / f, "rb").read(), f
    offs = SegmentMeta.load(str(tmp_path / "a" / "sn" / "v_0.meta")).blockOffsets
    dat = open(tmp_path / "a" / "sn" / "v_0.dat", "rb").read()
    got = b"".join(oracle_c.snappy_block_decode(dat[offs[k]:offs[k + 1]]) for k in range(len(offs) - 1))
    assert np.frombuffer(got, dtype="<i4").tolist() == [int(v) for v in vals]


def test_scala_set_order_building_blocks_are_pinned():
    """Engine.getColumns' order beyond 4 distinct columns is a Scala 2.12 immutable.HashSet's (Engine.scala:105).  The
    restatement (immutable3_amd/scala_sets.py) is pinned where it can be: MurmurHash3's mix / finalizeHash against an
    independent murmur3_32 (sklearn), String.hashCode against known values, and the hash-trie iteration order against
    scala-library 2.12's well-known outputs Set(1 to 5) -> (5, 1, 2, 3, 4) and Set(1 to 10) -> (5, 10, 1, 6, 9, 2, 7, 3, 8, 4)."""
    import struct
    from immutable3_amd import scala_sets as S
    assert S.java_string_hash("Map") == 77116 and S.java_string_hash("hello") == 99162322 and S.java_string_hash("") == 0
    try:
        from sklearn.utils import murmurhash3_32
    except Exception:
        murmurhash3_32 = None
    if murmurhash3_32 is not None:
        for seed in (0, 1, 0xcafebabe, 77116):
            for k in (0, 1, 0x87654321, 0xffffffff, 12345):
                assert murmurhash3_32(struct.pack("<I", k), seed=seed, positive=True) == S.finalize_hash(S.mix(seed, k), 4)
                assert murmurhash3_32(struct.pack("<II", k, k ^ 0x5bd1e995), seed=seed, positive=True) == \
                    S.finalize_hash(S.mix(S.mix(seed, k), k ^ 0x5bd1e995), 8)
    for n, want in ((4, [1, 2, 3, 4]), (5, [5, 1, 2, 3, 4]), (10, [5, 10, 1, 6, 9, 2, 7, 3, 8, 4])):
        st = S.ScalaSet()
        for i in range(1, n + 1):
            st.add(i, i)                   # Int.## is the int itself
        st.add(1, 1)                       # adding an element again changes nothing
        assert st.to_list() == want


def test_get_columns_order_small_sets_and_hash_sets(tmp_path):
    """<= 4 distinct columns: first-seen order (Set1..Set4).  >= 5: the HashSet's order -- the Python and the C++ planner
    must agree on it, and it is NOT first-seen order."""
    import subprocess
    from immutable3_amd.build import build_native
    names = ["id", "state", "age", "zip", "score", "flag", "tag"]
    cols = [Column.make("id", CodecType.DENSE_INT), Column.make("state", CodecType.DENSE_STRING, {"size": "2"}),
            Column.make("age", CodecType.DENSE_TINYINT), Column.make("zip", CodecType.DENSE_INT), Column.make("score", CodecType.DENSE_TINYINT),
            Column.make("flag", CodecType.DENSE_TINYINT), Column.make("tag", CodecType.DENSE_STRING, {"size": "3"})]
    t = Table("wide", cols, 1024)
    TableIO.store(str(tmp_path), t)
    small = Query("wide", And(Select("age", GT(1.0)), Select("zip", LT(5.0))), Project(["id", "age"], 0))
    assert [c.name for c in getColumns(small, t)] == ["age", "zip", "id"]
    big = Query("wide", And(And(Select("age", GT(1.0)), Select("zip", LT(5.0))), Select("state", Match(["CA"]))),
                Project(["id", "score", "flag", "tag", "age"], 0))
    order = [c.name for c in getColumns(big, t)]
    assert sorted(order) == sorted(names) and order != ["age", "zip", "state", "id", "score", "flag", "tag"]
    build_native()
    sql = "select id, score, flag, tag, age from wide where ((age > 1 and zip < 5) and state = 'CA')"
    p = subprocess.run([os.path.join(ROOT, "immutable3_amd", "bin", "imm3_sql"), "--parse-only", "-q", sql, "-d", str(tmp_path)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    used = [l for l in p.stdout.splitlines() if l.startswith("usedColumns:")][0].split()[1:]
    assert used == order, (used, order)


def test_the_single_pass_kernel_instances_that_count_their_own_loads_do_not_spill():
    """k_filter_project's streamers keep two tiles of loads in flight where registers allow (csrc/imm3_project.hip, project_depth):
    those instances issue their tile loads as inline asm and wait for them with s_waitcnt immediates of their own, because the
    compiler's wait-count analysis would drain the tile that should stay in flight.  Scratch traffic (a spill) would sit in the same
    counter and could store a register whose load has not landed: such an instance must fit its 168 registers.  The build keeps the
    compiler's resource remarks of that file (csrc/Makefile); instances that stay one tile ahead use the compiler's own waits and may
    spill a few registers, as they did in round 3."""
    import re
    from immutable3_amd.build import build_native
    build_native()
    path = os.path.join(ROOT, "immutable3_amd", "lib", "imm3_project.resources.txt")
    assert os.path.exists(path), "make -C immutable3_amd/csrc writes it next to the library"
    text = open(path).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
    assert len(blocks) == 16, len(blocks)                      # the 16 kind combinations of IMM3_PROJECT_KINDS
    set_regs = {0: 16, 1: 4, 2: 8, 3: 0}                       # TK_I32, TK_I8, TK_S2, TK_NONE (imm3_project.hip: tile_set_regs)
    seen_deep = 0
    for b in blocks:
        m = re.match(r"_ZN4imm316k_filter_projectILi(\d)ELi(\d)ELi(\d)E", b)
        assert m, b[:80]
        kinds = [int(x) for x in m.groups()]
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        assert vgprs <= 168, (kinds, vgprs)                    # 12 waves of a work-group = 3 per SIMD
        if sum(set_regs[k] for k in kinds) <= 20:              # two tiles ahead: untracked loads, hand-counted waits
            seen_deep += 1
            assert scratch == 0, (kinds, vgprs, scratch)
    assert seen_deep >= 9


def test_the_planners_cost_model_is_the_same_in_cpp_and_python():
    """csrc/imm3_plan.h (what the library decides with) and immutable3_amd/plan_model.py (what tools/plan_fit.py fits) are one model:
    same features, same coefficients (the header tools/plan_fit.py writes), on a grid of shapes, sizes and densities.  And the model
    says what the sweep it was fitted to says at its corners (profiles/r04_plan_sweep*): a small segment takes the bitmap path, C3's
    shape the one launch at 10 % survivors and three launches at 99 %, a sorted key's run the one launch."""
    from immutable3_amd import native, plan_model
    from immutable3_amd.build import build_native
    build_native()
    coef = plan_model.coefficients()
    shapes = [([(1, 0)], [(1, True)], 4), ([(1, 0), (4, 0)], [(4, True), (1, True)], 8), ([(2, 3)], [(4, False), (2, True), (1, False)], 4),
              ([(1, 0)], [(4, False)], 4), ([(4, 0)], [(4, True), (4, True)], 8), ([(2, 8), (1, 0)], [(2, True)], 4)]
    for pred, proj, rec in shapes:
        for n in (100_000, 4_000_000, 100_000_000):
            for sigma, sloc, full in ((0.0, 0.0, 0.0), (0.01, 0.01, 0.0), (0.1, 1.0, 1.0), (0.5, 0.5, 0.0), (0.99, 1.0, 0.3), (1.0, 1.0, 1.0)):
                got = native.plan_predict(n, pred, proj, rec, sigma, sloc, full)
                for plan in "ABC":
                    want = plan_model.cost(plan, n, sigma, sloc, full, pred, proj, rec, coef)
                    assert abs(got[plan] - want) <= 1e-6 * max(1.0, abs(want)), (plan, pred, proj, n, sigma, sloc, full, got[plan], want)
    c3 = ([(4, 0), (1, 0)], [(4, True), (1, True)], 8)
    def best(shape, n, sigma, sloc, full):
        c = native.plan_predict(n, shape[0], shape[1], shape[2], sigma, sloc, full)
        return min(c, key=c.get)
    assert best(c3, 100_000_000, 0.10, 0.10, 0.0) == "A"
    assert best(c3, 100_000_000, 0.99, 0.99, 0.0) == "C"
    assert best(c3, 4_000_000, 0.10, 0.10, 0.0) == "C"
    assert best(([(4, 0)], [(4, True)], 8), 100_000_000, 0.5, 1.0, 1.0) == "A"


def test_limit_scan_decision_is_a_named_function():
    """csrc/imm3_planner.cpp::limit_scan_applies (round 4 had it as an eleven-term boolean inside run_select): a projection with a
    `limit` scans in chunks that stop at the limit (Project.scala:73-80) exactly when its select chain is ONE tile launch over one
    uniform segment, a projection follows, and nothing wants the whole segment's count."""
    L = native.load()
    L.imm3_plan_limit_scan.restype = ctypes.c_int
    L.imm3_plan_limit_scan.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_int64] + [ctypes.c_int32] * 6 + [ctypes.c_int64]
    base = dict(whole=0, count_log_on=0, count_in_scan=1, limit=10, single_tile_pass=1, table=0, records=0, skip_bitmap=0, overlap_total=0, filter_variant=0, n_tiles=97657)
    call = lambda **kw: L.imm3_plan_limit_scan(*[{**base, **kw}[k] for k in base])
    assert call() == 1
    for veto in (dict(whole=1), dict(count_log_on=1), dict(count_in_scan=0), dict(limit=0), dict(limit=-3), dict(single_tile_pass=0), dict(table=1), dict(records=1),
                 dict(skip_bitmap=1), dict(overlap_total=1), dict(filter_variant=7), dict(filter_variant=14), dict(n_tiles=1024), dict(n_tiles=5)):
        assert call(**veto) == 0, veto
    assert call(n_tiles=1025) == 1 and call(filter_variant=12) == 1 and call(limit=10 ** 9) == 1


def test_bench_traffic_entries_are_stamped_with_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py only reports counter traffic (roofline.traffic, traffic_ratio, frac_traffic of C3 / C4 / C5 / the README-shaped table)
    from a profiles/traffic.json entry that was measured on THESE kernel sources; anything else is refused with the reason."""
    import json
    import bench
    sha = bench.extra_source_sha16()
    assert re.fullmatch(r"[0-9a-f]{16}", sha) and sha == bench.extra_source_sha16()
    good = {"extra": {"c3_range_age_id_project": {"kernels": ["k"], "hbm_bytes_per_query": 6.1e8, "source_sha16": sha, "tag": "t"},
                      "c5_table": {"kernels": ["k"], "hbm_bytes_per_pass": 5.0e9, "source_sha16": "0" * 16, "tag": "t"}}}
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps(good))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "extra_source_sha16", lambda: sha)
    e = bench.stamped_traffic("c3_range_age_id_project")
    assert e["hbm_bytes_per_query"] == 6.1e8 and "hash matches" in e["source"]
    stale = bench.stamped_traffic("c5_table")
    assert "hbm_bytes_per_pass" not in stale and "REFUSED" in stale["source"]
    assert "no entry" in bench.stamped_traffic("readme_table_c3")["source"]
    (tmp_path / "profiles" / "traffic.json").unlink()
    assert "missing" in bench.stamped_traffic("c3_range_age_id_project")["source"]
