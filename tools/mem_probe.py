"""Development tool: HBM write and copy rates with torch's own kernels (context for the store-heavy kernels)."""
import time, torch
x = torch.empty(100_000_000, dtype=torch.int32, device="cuda"); y = torch.empty_like(x)
xs = [torch.empty_like(x) for _ in range(3)]
def timed(f, n=30):
    for _ in range(3): f(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): f(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
t = timed(lambda i: xs[i % 3].fill_(7)); print(f"fill 400 MB: {t*1e6:7.1f} us  {0.4/t/1e3:6.2f} TB/s written")
t = timed(lambda i: xs[i % 3].copy_(xs[(i + 1) % 3])); print(f"copy 400 MB: {t*1e6:7.1f} us  {0.8/t/1e3:6.2f} TB/s read+written")
t = timed(lambda i: xs[i % 3].sum()); print(f"sum  400 MB: {t*1e6:7.1f} us  {0.4/t/1e3:6.2f} TB/s read")
b = torch.empty(40_000_000, dtype=torch.int8, device="cuda")
t = timed(lambda i: b.fill_(1)); print(f"fill 40 MB int8: {t*1e6:7.1f} us  {0.04/t/1e3:6.2f} TB/s")
