"""exploration (not a test): the benchFFT-style sweep of power-of-two and cubic c2c sizes;
prints ms per 4 GiB of data moved algorithmically (in + out) and the plan"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, torch
import fftw3_amd as fa


def run(label, shape, total=1 << 27):
    n = 1
    for v in shape:
        n *= v
    hm = max(1, total // n)
    x = torch.view_as_complex(torch.rand((hm * n, 2), dtype=torch.float64, device="cuda") - 0.5)
    y = torch.zeros_like(x)
    p = fa.plan_many_dft(len(shape), list(shape), hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    for _ in range(2):
        p.execute()
    p.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        p.execute()
    p.sync()
    dt = (time.perf_counter() - t0) / 5
    gb = 32.0 * n * hm / 1e9
    import math
    gf = 5.0 * n * math.log2(n) * hm / dt / 1e9
    trips = len(p.steps())
    print("%-14s x%-8d %7.3f ms %6.0f GB/s alg %6.0f GFLOPS  trips=%d  %s" % (
        label, hm, dt * 1e3, gb / dt, gf, trips, " ".join(l.strip().split(" ")[0] for l in p.sprint().split("\n")[1:])), flush=True)


lo, hi = (int(v) for v in os.environ.get("SWEEP_1D", "4,26").split(","))
for rep in range(int(os.environ.get("SWEEP_REPS", "1"))):
    for k in range(lo, hi + 1):
        run("1d 2^%d" % k, (1 << k,))
if os.environ.get("SWEEP_1D"):
    sys.exit(0)
for k in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
    run("2d %d^2" % k, (k, k))
for k in (32, 64, 128, 256, 512):
    run("3d %d^3" % k, (k, k, k))
