"""exploration: short transforms in large batches (not a test)"""
import time, torch
import fftw3_amd as fa
for n in (8, 16, 20, 25, 30, 32, 36, 45, 64, 100, 128, 256, 512, 1024):
    hm = (1 << 26) // n
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
    print("n=%-5d x%-8d %7.3f ms  %6.0f GB/s (%4.1f%%)  %s" % (n, hm, dt * 1e3, gb / dt, gb / dt / 80, p.sprint().replace("\n", " ")[:120]), flush=True)
