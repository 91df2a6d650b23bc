"""exploration: short r2r rows with the fused kernels (not a test)"""
import os, time, torch
import fftw3_amd as fa
NAMES = ["R2HC", "HC2R", "DHT", "REDFT00", "REDFT01", "REDFT10", "REDFT11", "RODFT00", "RODFT01", "RODFT10", "RODFT11"]
print("NO_R2CROWS =", os.environ.get("FFTW_AMD_NO_R2CROWS"))
for k, n in [(0, 1024), (1, 1024), (2, 1024), (5, 1024), (4, 1024), (3, 1025), (7, 1023), (6, 1024), (5, 256), (5, 2048)]:
    hm = (1 << 27) // n
    x = torch.rand(hm * n, dtype=torch.float64, device="cuda") - 0.5
    y = torch.zeros_like(x)
    p = fa.plan_many_r2r(1, [n], hm, x, None, 1, n, y, None, 1, n, [k])
    for _ in range(2): p.execute()
    p.sync()
    t0 = time.perf_counter()
    for _ in range(5): p.execute()
    p.sync()
    dt = (time.perf_counter() - t0) / 5
    gb = 16.0 * n * hm / 1e9
    print("%-8s n=%-5d %7.3f ms  %6.0f GB/s (in+out)  %s" % (NAMES[k], n, dt * 1e3, gb / dt, p.sprint().replace("\n", " ")[30:200]), flush=True)
