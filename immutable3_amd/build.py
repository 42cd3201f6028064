"""Builds lib/libimm3.so (hipcc, gfx950) from csrc/, the tools' build lib/libimm3_ablate.so (the same sources with the kernels'
ablation switches and the single-pass kernel's fault injection compiled in: tools/ and tests/test_gpu_fault_injection.py) and the
C++ host CLIs (bin/imm3_sql, bin/imm3_loader).  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))


def build_native(force: bool = False) -> str:
    csrc = os.path.join(_PKG, "csrc")
    out = os.path.join(_PKG, "lib", "libimm3.so")
    # every file the Makefile's rules depend on: an edit to any of them rebuilds under force=False
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".h")) or f == "Makefile"]
    inc = os.path.join(_PKG, "..", "include")
    srcs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    host = os.path.join(_PKG, "host")
    srcs += [os.path.join(host, f) for f in os.listdir(host) if f.endswith(".hpp")]
    srcs += [os.path.join(host, "cli", f) for f in os.listdir(os.path.join(host, "cli"))]
    outs = [out, os.path.join(_PKG, "lib", "libimm3_ablate.so"), os.path.join(_PKG, "bin", "imm3_sql"), os.path.join(_PKG, "bin", "imm3_loader")]
    if force or not all(os.path.exists(o) for o in outs) or min(os.path.getmtime(o) for o in outs) < max(os.path.getmtime(s) for s in srcs):
        jobs = max(1, min(8, os.cpu_count() or 1))   # one object per source file: the kernel files compile side by side
        cmd = ["make", "-C", csrc, "-s", f"-j{jobs}"] + (["-B"] if force else []) + ["all", "ablate"]
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_native(force=True))
