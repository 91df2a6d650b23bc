"""exploration (not a test): single huge power-of-two transforms checked by the shifted-impulse
known answer X[k] = exp(-2 pi i j0 k / n), evaluated on the device with exact index reduction"""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
for k in (26, 27, 28, 29):
    n = 1 << k
    j0 = 123456789 % n
    x = torch.zeros(n, dtype=torch.complex128, device="cuda")
    x[j0] = 1.0
    y = torch.zeros_like(x)
    p = fa.plan_dft_1d(n, x, y, fa.FORWARD)
    p.execute(); p.sync()
    t0 = time.perf_counter()
    p.execute(); p.sync()
    dt = time.perf_counter() - t0
    worst = 0.0
    step = 1 << 24
    for s in range(0, n, step):
        kk = torch.arange(s, min(n, s + step), dtype=torch.int64, device="cuda")
        m = (kk * j0) % n
        ang = m.to(torch.float64) * (-2.0 * math.pi / n)
        want = torch.complex(torch.cos(ang), torch.sin(ang))
        worst = max(worst, (y[s:s + step] - want).abs().max().item())
    print("2^%d: %.2f ms  %.0f GFLOPS  max abs err %.2e  %s" % (k, dt * 1e3, 5.0 * n * k / dt / 1e9, worst,
          [s.L for s in p.steps()]), flush=True)
    del p, x, y
