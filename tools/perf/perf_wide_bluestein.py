"""rows of 4097 ... 8192 points with a large prime factor: the one-kernel Bluestein on 512-item workgroups (round 3,
kernels_bluew.hip) against the step-by-step plans of round 2 (FFTW_AMD_NO_BLUE_ROWS=1: Rader / Bluestein in four or
five steps): ms per 4 GiB batch, whole % of the 8 TB/s roofline (32 bytes per point), error against torch.fft"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
x = torch.view_as_complex(torch.rand(((4 << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
print("%-8s %7s | %9s %7s %6s | %9s %7s %6s | %8s" % ("n", "howmany", "new ms", "whole%", "steps", "old ms", "whole%", "steps", "err"))
for n in (1031, 2053, 3001, 4093, 4099, 4513, 5003, 5006, 5501, 6007, 6521, 7001, 7499, 7919, 8191):
    hm = x.numel() // n
    res, steps = [], []
    err = 0.0
    for old in (0, 1):
        if old: os.environ["FFTW_AMD_NO_BLUE_ROWS"] = "1"
        else: os.environ.pop("FFTW_AMD_NO_BLUE_ROWS", None)
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
        p.execute(); p.sync()
        if not old:
            k = min(hm, 64)
            ref = torch.fft.fft(x[:k * n].reshape(k, n), dim=1)
            err = float((y[:k * n].reshape(k, n) - ref).abs().max() / ref.abs().max())
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        res.append(min(ts))
        steps.append(len(p.steps()))
        del p
    os.environ.pop("FFTW_AMD_NO_BLUE_ROWS", None)
    f = lambda t: 100 * 32.0 * n * hm / t / 8e12
    print("%-8d %7d | %9.3f %7.1f %6d | %9.3f %7.1f %6d | %8.1e" % (n, hm, res[0] * 1e3, f(res[0]), steps[0], res[1] * 1e3, f(res[1]), steps[1], err), flush=True)
