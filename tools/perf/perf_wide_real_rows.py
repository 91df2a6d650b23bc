"""real rows of 18 000 ... 30 720 points: the fused 512-item real-rows kernels (one trip, round 3) against the plans of
round 2 (FFTW_AMD_NO_R2CROWS=1: complex pass + untangle / tangle): ms per 2 GiB of reals, % of the roofline on n reals
in + (n/2+1) complex out"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
x = torch.rand((2 << 30) // 8, dtype=torch.float64, device="cuda") - 0.5
y = torch.zeros(x.numel() // 2 + (1 << 20), dtype=torch.complex128, device="cuda")
print("%-7s %-4s | %9s %7s | %9s %7s" % ("n", "kind", "new ms", "whole%", "old ms", "whole%"))
for n in (18000, 20000, 24576, 25600, 30720):
    hm = x.numel() // n
    h = n // 2 + 1
    byts = (8.0 * n + 16.0 * h) * hm
    for kind in ("r2c", "c2r"):
        res = []
        for old in (0, 1):
            if old: os.environ["FFTW_AMD_NO_R2CROWS"] = "1"
            else: os.environ.pop("FFTW_AMD_NO_R2CROWS", None)
            if kind == "r2c": p = fa.plan_many_dft_r2c(1, [n], hm, x, None, 1, n, y, None, 1, h)
            else: p = fa.plan_many_dft_c2r(1, [n], hm, y, None, 1, h, x, None, 1, n)
            p.execute(); p.sync()
            ts = []
            for _ in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
            res.append(min(ts)); del p
        os.environ.pop("FFTW_AMD_NO_R2CROWS", None)
        print("%-7d %-4s | %9.3f %7.1f | %9.3f %7.1f" % (n, kind, res[0] * 1e3, 100 * byts / res[0] / 8e12, res[1] * 1e3, 100 * byts / res[1] / 8e12), flush=True)
