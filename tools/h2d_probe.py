#!/usr/bin/env python3
"""Development tool: host -> HBM copy rates on this box (pageable, pinned, registered-in-place) for one 400 MB column."""
import time
import numpy as np
import torch

n = 400_000_000
a = np.random.default_rng(0).integers(0, 255, size=n, dtype=np.uint8)
t = torch.from_numpy(a)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()

def timed(f, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best

dt = timed(lambda: d.copy_(t))
print(f"pageable          : {n / dt / 1e9:6.1f} GB/s ({dt * 1e3:.1f} ms)")
t0 = time.perf_counter(); p = t.pin_memory(); pin_s = time.perf_counter() - t0
dt = timed(lambda: d.copy_(p, non_blocking=True))
print(f"pinned copy       : {n / dt / 1e9:6.1f} GB/s ({dt * 1e3:.1f} ms); pin_memory() itself {pin_s * 1e3:.1f} ms")
rt = torch.cuda.cudart()
t0 = time.perf_counter(); rc = rt.cudaHostRegister(t.data_ptr(), n, 0); reg_s = time.perf_counter() - t0
print(f"hipHostRegister   : rc={rc} {reg_s * 1e3:.1f} ms for 400 MB")
if int(rc) == 0:
    dt = timed(lambda: d.copy_(t, non_blocking=True))
    print(f"registered in place: {n / dt / 1e9:6.1f} GB/s ({dt * 1e3:.1f} ms)")
    h = n // 2
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def two():
        with torch.cuda.stream(s1): d[:h].copy_(t[:h], non_blocking=True)
        with torch.cuda.stream(s2): d[h:].copy_(t[h:], non_blocking=True)
    dt = timed(two)
    print(f"registered, 2 streams: {n / dt / 1e9:6.1f} GB/s ({dt * 1e3:.1f} ms)")
    t0 = time.perf_counter(); rt.cudaHostUnregister(t.data_ptr()); print(f"unregister {1e3 * (time.perf_counter() - t0):.1f} ms")
