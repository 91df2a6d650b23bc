"""contiguous rows of 1024 points: pass1024 (32 x 32, one exchange) against the three-stage rows kernel (4 x 16 x 16).
Run twice: FFTW_AMD_ROWS1024_3S=0 / 1 (the switch is read once per process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import fftw3_amd as fa
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from util import oracle_dft, aerror
n = 1024
rng = np.random.default_rng(5)
xs = (rng.random((37, n)) - 0.5) + 1j * (rng.random((37, n)) - 0.5)
for sign in (-1, 1):
    xd = torch.from_numpy(xs).cuda(); yd = torch.zeros_like(xd)
    p = fa.plan_many_dft(1, [n], 37, xd, None, 1, n, yd, None, 1, n, sign)
    p.execute(); p.sync()
    print("parity sign %d: %.2e" % (sign, aerror(yd.cpu().numpy(), oracle_dft(xs, (n,), 37, sign).reshape(37, n))))
x = torch.view_as_complex(torch.rand(((4 << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
hm = x.numel() // n
for inplace in (0, 1):
    p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, x if inplace else y, None, 1, n, -1)
    p.execute(); p.sync()
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
    print("ROWS1024_3S=%s inplace=%d: %.3f ms  %.1f%%" % (os.environ.get("FFTW_AMD_ROWS1024_3S", "0"), inplace, min(ts) * 1e3, 100 * 32.0 * n * hm / min(ts) / 8e12))
