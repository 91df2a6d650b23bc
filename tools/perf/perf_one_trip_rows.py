"""exploration (not a test): lengths that gained a one-trip three-stage rows kernel (13-smooth up to 4096, menu lengths in
(4096, 8192]) against what the planner did before (FFTW_AMD_NO_3S=1); 4 GiB of complex128 per case"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
tot = (4 << 30) // 16
x = (torch.rand(tot, dtype=torch.float64, device="cuda") - 0.5).to(torch.complex128)
y = torch.zeros_like(x)
for n in (1001, 1100, 1144, 1573, 2002, 3003, 4004, 4116, 4400, 5000, 5120, 6000, 6144, 6561, 7168, 7200, 7680, 8064):
    hm = tot // n
    row = []
    for no3s in (1, 0):
        if no3s: os.environ["FFTW_AMD_NO_3S"] = "1"
        else: os.environ.pop("FFTW_AMD_NO_3S", None)
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD, fa.ESTIMATE)
        p.execute(); p.sync()
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        row.append("%7.3f ms %6.0f GF whole %4.1f%% [%s]" % (t * 1e3, 5.0 * n * math.log2(n) * hm / t / 1e9, 100 * 32.0 * n * hm / t / 8e12,
                   " ".join(l.strip().split(" tile")[0].lstrip("(") for l in p.sprint().splitlines()[1:])))
        del p
    print("n=%-5d before: %s | now: %s" % (n, row[0], row[1]), flush=True)
