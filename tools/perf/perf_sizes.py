"""exploration: throughput of sizes without a register kernel (not a test)"""
import time, torch
import fftw3_amd as fa
for n in (1000, 1536, 1920, 2000, 2400, 3000, 3600, 4000, 5000, 10000, 60060, 100000, 360, 720, 1080):
    hm = max(1, (1 << 29) // (16 * n) * 2)          # ~1 GiB of input
    x = torch.view_as_complex(torch.rand((hm * n, 2), dtype=torch.float64, device="cuda") - 0.5)
    y = torch.zeros_like(x)
    p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    for _ in range(2): p.execute()
    p.sync()
    t0 = time.perf_counter()
    for _ in range(5): p.execute()
    p.sync()
    dt = (time.perf_counter() - t0) / 5
    gb = 32.0 * n * hm / 1e9
    print("n=%-8d x%-7d %7.3f ms  %6.0f GB/s alg (%4.1f%% of 8 TB/s)  %s" % (n, hm, dt * 1e3, gb / dt, gb / dt / 80, p.sprint().replace("\n", " ")[:200]), flush=True)
    del x, y, p
