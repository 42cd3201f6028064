"""Builds lib/libimm3.so (hipcc, gfx950) from csrc/.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))


def build_native(force: bool = False) -> str:
    csrc = os.path.join(_PKG, "csrc")
    out = os.path.join(_PKG, "lib", "libimm3.so")
    srcs = [os.path.join(csrc, f) for f in ("imm3_kernels.hip", "imm3_api.cpp", "imm3_internal.h")]
    srcs.append(os.path.join(_PKG, "..", "include", "imm3.h"))
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in srcs):
        cmd = ["make", "-C", csrc, "-s"] + (["-B"] if force else [])
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_native(force=True))
