"""r2c and c2r of long mixed lengths n = 4m whose quarter length is no power of two: the half-length plans FFTW_ESTIMATE
now picks against the radix-4 plans of round 2 (FFTW_AMD_FORCE_RADIX4=1, a pass on the LDS kernel): ms per 4 GiB"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
buf = torch.rand((4 << 30) // 8, dtype=torch.float64, device="cuda") - 0.5
for n in (3932160, 3145728, 4096000, 2457600, 3000000, 1200000, 6000000, 1 << 22):
    hm = buf.numel() // n
    x = buf[:hm * n].reshape(hm, n)
    y = torch.zeros(hm, n // 2 + 1, dtype=torch.complex128, device="cuda")
    xb = torch.zeros(hm, n, dtype=torch.float64, device="cuda")
    out = []
    for old in (0, 1):
        if old: os.environ["FFTW_AMD_FORCE_RADIX4"] = "1"
        else: os.environ.pop("FFTW_AMD_FORCE_RADIX4", None)
        p = fa.plan_many_dft_r2c(1, [n], hm, x, None, 1, n, y, None, 1, n // 2 + 1)
        q = fa.plan_many_dft_c2r(1, [n], hm, y, None, 1, n // 2 + 1, xb, None, 1, n)
        p.execute(); p.sync(); q.execute(); q.sync()
        err = float((xb[:2] / n - x[:2]).abs().max())
        tt = []
        for pl in (p, q):
            ts = []
            for _ in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter(); pl.execute(); pl.sync(); ts.append(time.perf_counter() - t0)
            tt.append(min(ts))
        p.execute(); p.sync()                       # c2r may destroy its input: restore the spectrum
        out.append((tt[0], tt[1], len(p.steps()), len(q.steps()), err))
        del p, q
    os.environ.pop("FFTW_AMD_FORCE_RADIX4", None)
    f = lambda t: 100 * 16.0 * n * hm / t / 8e12
    print("%8d x %-5d r2c %d steps %7.3f ms %5.1f %% (radix-4: %d steps %7.3f ms) | c2r %d steps %7.3f ms %5.1f %% (radix-4: %d steps %7.3f ms) round trip %.1e" %
          (n, hm, out[0][2], out[0][0] * 1e3, f(out[0][0]), out[1][2], out[1][0] * 1e3, out[0][3], out[0][1] * 1e3, f(out[0][1]), out[1][3], out[1][1] * 1e3, out[0][4]), flush=True)
    del y, xb
