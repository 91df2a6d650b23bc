"""rows of 8193 ... 16383 points: the wide three-stage kernels (one trip, 512-item workgroups, round 3) against the
two-pass plans of round 2 (FFTW_AMD_NO_3S=1): ms per 4 GiB batch, whole % of the 8 TB/s roofline"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
x = torch.view_as_complex(torch.rand(((4 << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
print("%-8s %7s | %9s %7s | %9s %7s" % ("n", "howmany", "new ms", "whole%", "old ms", "whole%"))
for n in (8400, 9000, 9216, 10000, 10240, 10800, 12000, 12288, 12800, 13824, 14400, 15360):
    hm = x.numel() // n
    res = []
    for old in (0, 1):
        if old: os.environ["FFTW_AMD_NO_3S"] = "1"
        else: os.environ.pop("FFTW_AMD_NO_3S", None)
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
        p.execute(); p.sync()
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        res.append(min(ts))
        del p
    os.environ.pop("FFTW_AMD_NO_3S", None)
    print("%-8d %7d | %9.3f %7.1f | %9.3f %7.1f" % (n, hm, res[0] * 1e3, 100 * 32.0 * n * hm / res[0] / 8e12, res[1] * 1e3, 100 * 32.0 * n * hm / res[1] / 8e12), flush=True)
