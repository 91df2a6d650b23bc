"""exploration (not a test): create / execute / destroy many plans and watch free device memory"""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import fftw3_amd as fa
rng = np.random.default_rng(0)
x = torch.randn(1 << 22, dtype=torch.complex128, device="cuda")
y = torch.zeros_like(x)
xr = torch.randn(1 << 22, dtype=torch.float64, device="cuda")
yr = torch.zeros_like(xr)
yc = torch.zeros((1 << 21) + 4096, dtype=torch.complex128, device="cuda")
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
sizes = [1000, 1024, 4096, 15015, 1 << 18, 9973, 3000, 77, 2 * 3 * 5 * 7 * 11 * 13]
for it in range(600):
    n = sizes[it % len(sizes)]
    hm = max(1, min(64, (1 << 21) // n))
    kind = it % 4
    if kind == 0:
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    elif kind == 1:
        p = fa.plan_many_dft_r2c(1, [n], hm, xr, None, 1, n, yc, None, 1, n // 2 + 1)
    elif kind == 2:
        p = fa.plan_many_r2r(1, [n], hm, xr, None, 1, n, yr, None, 1, n, [fa.REDFT10])
    else:
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, x, None, 1, n, fa.BACKWARD)
    p.execute()
    p.sync()
    del p
    if it % 100 == 99:
        gc.collect()
        torch.cuda.synchronize()
        print(it + 1, "plans; free device memory change: %.1f MiB" % ((torch.cuda.mem_get_info()[0] - free0) / 2**20), flush=True)
