"""exploration: REDFT10 2^20 batch 1024 at several chunk sizes (not a test)"""
import time, numpy as np, torch
import fftw3_amd as fa
n, hm = 1 << 20, 1024
x = torch.rand(hm * n, dtype=torch.float64, device="cuda") - 0.5
y = torch.zeros_like(x)
for cb in (256 << 20, 1 << 30, 2 << 30, 4 << 30, 8 << 30):
    fa.set_chunk_bytes(cb)
    p = fa.plan_many_r2r(1, [n], hm, x, None, 1, n, y, None, 1, n, [fa.REDFT10])
    for _ in range(2): p.execute()
    p.sync()
    t0 = time.perf_counter()
    for _ in range(5): p.execute()
    p.sync()
    dt = (time.perf_counter() - t0) / 5
    prof = p.execute_profiled()
    print("chunk_bytes %5d MiB chunk=%4d  %.3f ms  %.2f us/transform   steps(ms/launch): %s" % (
        cb >> 20, p.chunk, dt * 1e3, dt / hm * 1e6, " ".join("%.3f" % (ms / c) for _, ms, c in prof)), flush=True)
    del p
# the r2c of the same length for comparison
fa.set_chunk_bytes(0)
z = torch.zeros(hm * (n // 2 + 1), dtype=torch.complex128, device="cuda")
p = fa.plan_many_dft_r2c(1, [n], hm, x, None, 1, n, z, None, 1, n // 2 + 1)
for _ in range(2): p.execute()
p.sync()
t0 = time.perf_counter()
for _ in range(5): p.execute()
p.sync()
dt = (time.perf_counter() - t0) / 5
prof = p.execute_profiled()
print("r2c chunk=%d %.3f ms %.2f us/transform steps: %s" % (p.chunk, dt * 1e3, dt / hm * 1e6, " ".join("%.3f" % (ms / c) for _, ms, c in prof)))
