"""exploration (not a test): 2^21 and 2^22 as TWO passes through a strided 2048-point kernel with 4-wide tiles
(64-byte segments) against the three-pass default"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, "tests")
import numpy as np, torch
import fftw3_amd as fa
from util import oracle_dft, aerror
x = torch.view_as_complex(torch.rand(((8 << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
for lg, forced in ((21, None), (21, "2048,1024"), (21, "1024,2048"), (22, None), (22, "2048,2048"), (23, None), (20, None)):
    n = 1 << lg
    hm = x.numel() // n
    if forced: os.environ["FFTW_AMD_FORCE_LENS"] = forced
    else: os.environ.pop("FFTW_AMD_FORCE_LENS", None)
    p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    p.execute(); p.sync()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    got = y[:2 * n].cpu().numpy().reshape(2, n)
    want = oracle_dft(x[:2 * n].cpu().numpy(), (n,), 2).reshape(2, n)
    prof = p.execute_profiled()
    print("2^%d %-10s %7.2f ms %5.0f GF %4.1f%%  err %.1e  steps %s  %s" % (lg, forced or "default", t * 1e3, 5.0 * n * lg * hm / t / 1e9,
          100 * 32.0 * n * hm / t / 8e12, aerror(got, want), [round(m, 2) for _, m, l in prof],
          " ".join(l.strip().split(" buf")[0] for l in p.sprint().splitlines()[1:])), flush=True)
    del p
