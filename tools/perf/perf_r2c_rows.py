"""exploration (not a test): batched r2c / c2r of mid-size rows, 2 GiB of reals per case"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
tot = (2 << 30) // 8
x = torch.rand(tot, dtype=torch.float64, device="cuda") - 0.5
SIZES = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384, 32768, 1000, 1920, 3000, 4000, 6000, 10000]
# the half spectra of all rows: (n / 2 + 1) / n of the real size, largest for the shortest rows
z = torch.zeros(max((tot // n) * (n // 2 + 1) for n in SIZES) + 1024, dtype=torch.complex128, device="cuda")
for n in SIZES:
    hm = tot // n
    for kind in ("r2c", "c2r"):
        if kind == "r2c": p = fa.plan_many_dft_r2c(1, [n], hm, x, None, 1, n, z, None, 1, n // 2 + 1)
        else: p = fa.plan_many_dft_c2r(1, [n], hm, z, None, 1, n // 2 + 1, x, None, 1, n)
        p.execute(); p.sync()
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        print("%s n=%-6d x%-7d %7.3f ms %6.0f GF whole %4.1f%%  %s" % (kind, n, hm, t * 1e3, 2.5 * n * math.log2(n) * hm / t / 1e9,
              100 * (8.0 * n + 16.0 * (n // 2 + 1)) * hm / t / 8e12,
              " ".join(l.strip().split(" tile")[0].split(" n=")[0].lstrip("(") for l in p.sprint().splitlines()[1:])), flush=True)
        del p
