import sys
sys.path.insert(0, "/root/repo")
import torch
from immutable3_amd import native
ctx = native.Context(0)
for nb in (100_000_000, 200_000_000, 400_000_000, 1_000_000_000):
    print(nb, [round(ctx.measure_read_gbps(nb, 30)) for _ in range(3)])
