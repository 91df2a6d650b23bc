"""exploration (not a test): batched 1-D r2c / c2r, 4 GiB of reals each"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
x = torch.rand((4 << 30) // 8, dtype=torch.float64, device="cuda") - 0.5
z = torch.zeros(x.numel() // 2 + (1 << 20), dtype=torch.complex128, device="cuda")
for lg in (14, 16, 18, 19, 20, 21, 22):
    n = 1 << lg
    hm = x.numel() // n
    for kind in ("r2c", "c2r"):
        if kind == "r2c": p = fa.plan_many_dft_r2c(1, [n], hm, x, None, 1, n, z, None, 1, n // 2 + 1)
        else: p = fa.plan_many_dft_c2r(1, [n], hm, z, None, 1, n // 2 + 1, x, None, 1, n)
        p.execute(); p.sync()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        prof = p.execute_profiled()
        print("%s 2^%-2d x%-7d %7.3f ms %6.0f GF %4.1f%%  steps %s  %s" % (kind, lg, hm, t * 1e3, 2.5 * n * lg * hm / t / 1e9,
              100 * (8.0 * n + 16.0 * (n // 2 + 1)) * hm / t / 8e12, [round(m, 2) for _, m, l in prof],
              " ".join(l.strip().split(" tile")[0].split(" n=")[0].lstrip("(") for l in p.sprint().splitlines()[1:])), flush=True)
        del p
