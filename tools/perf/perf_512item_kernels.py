import math, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import fftw3_amd as fa
x = torch.view_as_complex(torch.rand(((4 << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
for shape in ((16384,), (8192,), (4096, 4096), (2048, 4096)):
    n = 1
    for v in shape: n *= v
    hm = x.numel() // n
    for sign in (-1, 1):
        p = fa.plan_many_dft(len(shape), list(shape), hm, x, None, 1, n, y, None, 1, n, sign)
        p.execute(); p.sync()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        print("%-12s sign %2d %9.3f ms %7.1f%%" % ("x".join(str(v) for v in shape), sign, t * 1e3, 100 * 32.0 * n * hm / t / 8e12), flush=True)
