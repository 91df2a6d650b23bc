"""The north-star sweep on one GPU: powers of two 2^10 ... 2^24, 7-smooth and {11, 13}-containing lengths, the
primes 17 / 1031 / 65537 (O(p^2) stage, Bluestein, Rader), some 2-D / 3-D shapes; batched c2c, forward, out of
place, FFTW_ESTIMATE, about 4 GiB of input each.  GFLOPS = 5 N log2 N * howmany / t, whole % = 32 N howmany / t
against 8 TB/s.  -> profiles/r0x_sweep.txt (r03: with chunk lanes)"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
gib = int(os.environ.get("GIB", "4"))
x = torch.view_as_complex(torch.rand(((gib << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
shapes = [(n,) for n in (8, 16, 17, 32, 64, 100, 256, 512)] + [(1 << k,) for k in range(10, 25)] + [(n,) for n in (
    1000, 1080, 1920, 2000, 3000, 3600, 5000, 10000, 15015, 30030, 60060, 100000, 250000, 518400, 1000000, 1105920,
    2000000, 10000000, 15375360, 143, 1001, 1100, 4004, 6000, 1031, 65537, 17408)] + [
    (1024, 1024), (1080, 1920), (2048, 2048), (4096, 4096), (128, 128, 128), (256, 256, 256), (100, 100, 100)]
print("%-18s %9s %9s %8s %7s  %s" % ("shape", "howmany", "ms", "GFLOPS", "whole%", "plan"))
for shape in shapes:
    n = 1
    for v in shape: n *= v
    hm = max(1, x.numel() // n)
    try:
        p = fa.plan_many_dft(len(shape), list(shape), hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    except Exception as e:
        print(shape, "no plan", e); continue
    p.execute(); p.sync()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    plan = " ".join(l.strip().split(" tile")[0].split(" buf")[0].lstrip("(") for l in p.sprint().splitlines()[1:])
    print("%-18s %9d %9.3f %8.0f %7.1f  %s" % ("x".join(str(v) for v in shape), hm, t * 1e3, 5.0 * n * math.log2(n) * hm / t / 1e9,
                                               100 * 32.0 * n * hm / t / 8e12, plan), flush=True)
    del p
