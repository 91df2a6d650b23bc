"""strided axes of 1025 ... 2048 points: the 512-item strided kernels (8 ... 15 sequences per tile, round 3) -- timing of
the shapes that use them; compare with profiles/r03_sweep.txt / r02 numbers (2^21: 29.8 %, 1080 x 1920: 22.0 %)"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
x = torch.view_as_complex(torch.rand(((4 << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
for shape in ((1 << 21,), (1080, 1920), (1200, 1600), (2048, 1024), (1440, 2560), (2000, 1000), (1536, 1536), (2048, 8192), (1105920,), (2000000,)):
    n = 1
    for v in shape: n *= v
    hm = x.numel() // n
    p = fa.plan_many_dft(len(shape), list(shape), hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    p.execute(); p.sync()
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    plan = " ".join(l.strip().split(" buf")[0].lstrip("(") for l in p.sprint().splitlines()[1:])
    print("%-12s %6d %9.3f ms %6.1f%%  %s" % ("x".join(str(v) for v in shape), hm, t * 1e3, 100 * 32.0 * n * hm / t / 8e12, plan), flush=True)
    del p
