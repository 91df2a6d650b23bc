"""exploration: per-step time of r2r plans (not a test)"""
import sys, time
import numpy as np, torch
import fftw3_amd as fa
NAMES = ["R2HC", "HC2R", "DHT", "REDFT00", "REDFT01", "REDFT10", "REDFT11", "RODFT00", "RODFT01", "RODFT10", "RODFT11"]
def run(shape, kinds, hm, label):
    n = int(np.prod(shape))
    x = torch.rand(hm * n, dtype=torch.float64, device="cuda") - 0.5
    y = torch.zeros_like(x)
    p = fa.plan_many_r2r(len(shape), shape, hm, x, None, 1, n, y, None, 1, n, kinds)
    for _ in range(3): p.execute()
    p.sync()
    t0 = time.perf_counter()
    for _ in range(10): p.execute()
    p.sync()
    dt = (time.perf_counter() - t0) / 10
    gb = 2 * 8 * hm * n / 1e9
    print("%-22s %8.3f ms  %7.1f GB/s (in+out)" % (label, dt * 1e3, gb / dt), flush=True)
    prof = p.execute_profiled()
    for s, ms, cnt in prof:
        print("      step kind=%d var=%d L=%d  %.3f ms x%d" % (s.kind, s.variant, s.L, ms, cnt))
if __name__ == "__main__":
    for k in range(11):
        n = (1 << 20) + (1 if k == 3 else (-1 if k == 7 else 0))
        run([n], [k], 64, NAMES[k] + " %d x64" % n)
    run([4096, 4096], [5, 5], 4, "DCT2 2D 4096^2 x4")
    run([1000], [5], 65536, "DCT2 1000 x65536")
    run([4095], [7], 16384, "DST1 4095 x16384")
