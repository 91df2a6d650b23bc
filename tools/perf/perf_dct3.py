"""exploration (not a test): REDFT01 / RODFT01 / REDFT10 / RODFT10 of 2^20 points x 2048, per-step times"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
n, b = 1 << 20, 2048
x = torch.rand(b * n, dtype=torch.float64, device="cuda") - 0.5
y = torch.zeros_like(x)
for name, kind in (("REDFT10", fa.REDFT10), ("RODFT10", fa.RODFT10), ("REDFT01", fa.REDFT01), ("RODFT01", fa.RODFT01)):
    p = fa.plan_many_r2r(1, [n], b, x, None, 1, n, y, None, 1, n, [kind])
    p.execute(); p.sync()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
    prof = p.execute_profiled()
    print("%s %.2f ms  steps %s" % (name, min(ts) * 1e3, [round(m, 2) for _, m, l in prof]), flush=True)
