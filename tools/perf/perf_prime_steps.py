"""per-step times of the Rader / Bluestein plans beyond one kernel (where do the 8 steps spend their time?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
x = torch.view_as_complex(torch.rand(((2 << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
for n in (65537, 12289, 8191, 40961, 10007, 100003):
    hm = x.numel() // n
    p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    p.execute(); p.sync()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
    prof = p.execute_profiled()
    names = [l.strip().split(" tile")[0].split(" buf")[0].lstrip("(") for l in p.sprint().splitlines()[1:]]
    print("n=%d howmany=%d  %.3f ms  whole %.1f%%  chunk=%d lanes=%d" % (n, hm, min(ts) * 1e3, 100 * 32.0 * n * hm / min(ts) / 8e12, p.chunk, p.lanes))
    for nm, t in zip(names, prof):
        print("    %-34s %8.3f ms  (%d launches)" % (nm, t[1], t[2]))
    del p
