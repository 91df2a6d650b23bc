"""exploration (not a test): lengths with a large prime factor, the one-kernel Bluestein (pass3b.hpp) against the
step-by-step plans (FFTW_AMD_NO_BLUE_ROWS=1: Bluestein in five steps, or Rader); 2 GiB of complex128 per case"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
tot = (2 << 30) // 16
x = (torch.rand(tot, dtype=torch.float64, device="cuda") - 0.5).to(torch.complex128)
y = torch.zeros_like(x)
for n in (211, 269, 331, 509, 1009, 1018, 1031, 2053, 3001, 3833):
    hm = tot // n
    row = []
    for off in ("1", ""):
        if off: os.environ["FFTW_AMD_NO_BLUE_ROWS"] = "1"
        else: os.environ.pop("FFTW_AMD_NO_BLUE_ROWS", None)
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD, fa.ESTIMATE)
        p.execute(); p.sync()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        row.append("%8.3f ms %5.0f GF whole %4.1f%% (%d steps)" % (t * 1e3, 5.0 * n * math.log2(n) * hm / t / 1e9, 100 * 32.0 * n * hm / t / 8e12, len(p.steps())))
        del p
    print("n=%-5d steps: %s | one kernel: %s" % (n, row[0], row[1]), flush=True)
