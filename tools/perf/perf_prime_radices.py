"""exploration (not a test): lengths with factors 17 / 19 / 23, register kernels against the runtime-radix LDS kernel
(FFTW_AMD_NO_TUNED=1); 2 GiB of complex128 per case"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
tot = (2 << 30) // 16
x = torch.rand(tot, dtype=torch.float64, device="cuda").to(torch.complex128)
y = torch.zeros_like(x)
for n in (34, 136, 272, 289, 304, 323, 368, 17408, 18496, 19456, 136 * 1024, 323 * 1024):
    hm = tot // n
    row = []
    for tuned in (0, 1):
        if tuned: os.environ.pop("FFTW_AMD_NO_TUNED", None)
        else: os.environ["FFTW_AMD_NO_TUNED"] = "1"
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD, fa.ESTIMATE)
        p.execute(); p.sync()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        row.append("%8.3f ms %5.2f TB/s/pass %s" % (t * 1e3, 32.0 * n * hm / t / 1e12,
                   [(s.L, s.variant) for s in p.steps()]))
        del p
    print("n=%-7d x%-8d  lds: %s | reg: %s" % (n, hm, row[0], row[1]), flush=True)
