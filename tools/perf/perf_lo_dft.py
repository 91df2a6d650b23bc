"""2-D shapes whose strided axis is split T x L0 and finished by the rows kernel (FFTW_AMD_F_LO_DFT, round 3) against
the round-2 plans (FFTW_AMD_NO_LO_DFT=1): ms per 4 GiB batch, whole % of the 8 TB/s roofline on 32 N bytes."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
x = torch.view_as_complex(torch.rand(((4 << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
print("%-12s %5s | %9s %7s | %9s %7s | plan" % ("shape", "hm", "new ms", "whole%", "old ms", "whole%"))
for n0, n1 in ((4096, 4096), (2048, 4096), (4096, 2048), (2048, 2048), (1536, 2048), (1280, 4096), (3072, 4096)):
    n = n0 * n1
    hm = x.numel() // n
    res = []
    for old in (0, 1):
        if old: os.environ["FFTW_AMD_NO_LO_DFT"] = "1"
        else: os.environ.pop("FFTW_AMD_NO_LO_DFT", None)
        p = fa.plan_many_dft(2, [n0, n1], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
        p.execute(); p.sync()
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        res.append((min(ts), " ".join(l.strip().split(" tile")[0].lstrip("(") for l in p.sprint().splitlines()[1:])))
        del p
    os.environ.pop("FFTW_AMD_NO_LO_DFT", None)
    print("%-12s %5d | %9.3f %7.1f | %9.3f %7.1f | %s  <-  %s" % ("%dx%d" % (n0, n1), hm, res[0][0] * 1e3, 100 * 32.0 * n * hm / res[0][0] / 8e12,
          res[1][0] * 1e3, 100 * 32.0 * n * hm / res[1][0] / 8e12, res[0][1], res[1][1]), flush=True)
