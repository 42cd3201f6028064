"""The host-side concurrency pieces of the C ABI (immutable3_amd/csrc/imm3_sync.h: per-context buffer pool, capture gate,
handle reference counts, slot counter) under -fsanitize=thread, driven by 8 threads the way the reference drives the
path (FixedThreadPool(cpuCount), one PipelineThread per segment: Engine.scala:176-180,247-262; SqlCli.scala:64).
No GPU needed: the pool's backend is malloc / free.  The GPU half of the contract is tests/test_gpu_threads.py."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "tsan_sync.cpp")


def _build(tmp_path, name, flags):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", *flags, SRC, "-o", exe])
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_sync_primitives_plain_build(tmp_path):
    exe = _build(tmp_path, "sync_plain", [])
    r = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_sync_primitives_under_thread_sanitizer(tmp_path):
    # the sanitizer is live in this image: a planted race is reported ...
    racy = _build(tmp_path, "sync_racy", ["-fsanitize=thread", "-DPLANT_RACE"])
    r = subprocess.run([racy, "200"], capture_output=True, text=True, timeout=300)
    if "ThreadSanitizer" not in r.stderr and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory on this kernel")
    assert "WARNING: ThreadSanitizer: data race" in r.stderr
    # ... and the real pieces are clean
    exe = _build(tmp_path, "sync_tsan", ["-fsanitize=thread"])
    r = subprocess.run([exe, "5000"], capture_output=True, text=True, timeout=600, env={**os.environ, "TSAN_OPTIONS": "halt_on_error=1"})
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.stdout.strip().endswith("ok")
