"""fused two-pass kernel: correctness vs the two-launch path and the oracle, then timing"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import fftw3_amd as fa
from util import oracle_dft, aerror
dev = torch.device("cuda:0")
n = 1 << 20
ok = True
for b in (8, 37, 64):
    x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.zeros_like(x)
    for sign in (-1, 1):
        p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, sign)
        for rep in range(3):
            y.zero_(); p.execute(); torch.cuda.synchronize()
            ref = torch.fft.fft(x, dim=1) if sign < 0 else torch.fft.ifft(x, dim=1) * n
            e = float((y - ref).abs().max() / ref.abs().max())
            rows = [0, b // 2, b - 1]
            eo = aerror(y[rows].cpu().numpy(), oracle_dft(x[rows].cpu().numpy(), (n,), 3, sign).reshape(3, n))
            good = e < 1e-12 and eo < 1e-10
            ok &= good
            print("b=%d sign=%d rep=%d: vs torch.fft %.2e  vs oracle %.2e %s" % (b, sign, rep, e, eo, "ok" if good else "FAIL"), flush=True)
# in place
x = torch.randn(16, n, dtype=torch.complex128, device=dev); x0 = x.clone()
p = fa.plan_many_dft(1, [n], 16, x, None, 1, n, x, None, 1, n, -1); p.execute(); torch.cuda.synchronize()
e = float((x - torch.fft.fft(x0, dim=1)).abs().max() / x0.abs().max() / 1024)
print("in-place b=16: %.2e" % e); ok &= e < 1e-12
for b in (512, int(os.environ.get("BIG", "2048"))):
    x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
    p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1)
    p.execute(); torch.cuda.synchronize()
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize(); t = time.perf_counter(); p.execute(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print("b=%d: %.3f ms  %.2f us/xform  %.0f GFLOPS  %.1f%% of roofline" % (b, best * 1e3, best / b * 1e6, 5 * n * 20 * b / best / 1e9, 32 * n * b / best / 8e12 * 100), flush=True)
    # spot check a few rows of the big run against torch
    idx = torch.tensor([0, b // 3, b - 1], device=dev)
    e = float((y[idx] - torch.fft.fft(x[idx], dim=1)).abs().max() / 1024)
    print("   spot check %.2e" % e); ok &= e < 1e-11
print("ALL OK" if ok else "SOME FAILED")
