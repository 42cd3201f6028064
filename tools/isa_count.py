#!/usr/bin/env python3
"""Development tool (no GPU needed): instruction mix of one kernel's loops, read from hipcc's assembly listing.

usage: isa_count.py <source.hip> <mangled-kernel-substring> [-D...]
Compiles the file for gfx950 with -save-temps, finds the kernel whose mangled name contains the substring, and prints for every loop
(LLVM annotates each block with its innermost loop header and depth) the number of VALU / SALU / LDS / VMEM / SMEM instructions in
the blocks that belong to it directly -- cold paths included, so the figures are an upper bound per iteration, good for comparing two
versions of the same source.  Also prints register use and the number of SGPR-spill lane moves (v_readlane / v_writelane)."""
import os
import re
import subprocess
import sys
import tempfile

src, needle = sys.argv[1], sys.argv[2]
flags = sys.argv[3:]
tmp = tempfile.mkdtemp(prefix="isa_")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-parameter", *flags, "-c", os.path.abspath(src),
                       "-o", os.path.join(tmp, "x.o"), "-save-temps=obj"], cwd=tmp)
lst = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
lines = open(os.path.join(tmp, lst)).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and needle in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
tail = next(i for i in range(end, len(lines)) if ".end_amdhsa_kernel" in lines[i])
meta = [l.strip() for l in lines[end:tail + 40] if re.search(r"NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|next_free_vgpr|sgpr_count", l)]


def kind(op):
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


loops = {}
cur = ("entry", 0)
for l in lines[start:end + 1]:
    m = re.match(r"^\.LBB\d+_\d+:", l)
    if m:
        cur = ("entry", 0)
        h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
        if h:
            cur = (h.group(1), int(h.group(2)))
        continue
    h = re.search(r"^\s*;\s+(?:in Loop|=>\s*This Inner Loop Header|This Inner Loop Header|Parent Loop|Child Loop).*", l)
    if h:
        m2 = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
        if m2:
            cur = (m2.group(1), int(m2.group(2)))
        m3 = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", l)
        if m3:
            cur = ("self@%d" % len(loops), int(m3.group(1)))
        continue
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        continue
    op = s.split()[0]
    d = loops.setdefault(cur, {})
    k = kind(op)
    d[k] = d.get(k, 0) + 1
    if op.startswith("s_waitcnt"):
        d["waits"] = d.get("waits", 0) + 1
    if op.startswith("s_nop"):
        d["nops"] = d.get("nops", 0) + 1
print("\n".join(meta))
print(f"{'loop':>14s} depth  total  valu  salu  lane   lds  vmem  smem  (waits nops)")
for (name, depth), d in sorted(loops.items(), key=lambda kv: -sum(v for k, v in kv[1].items() if k not in ("waits", "nops"))):
    tot = sum(v for k, v in d.items() if k not in ("waits", "nops"))
    if tot < 40:
        continue
    print(f"{name:>14s} {depth:5d} {tot:6d} {d.get('valu', 0):5d} {d.get('salu', 0):5d} {d.get('lane', 0):5d} {d.get('lds', 0):5d} {d.get('vmem', 0):5d} {d.get('smem', 0):5d}  ({d.get('waits', 0)} {d.get('nops', 0)})")
print("listing:", os.path.join(tmp, lst), "kernel lines", start + 1, "-", end + 1)
