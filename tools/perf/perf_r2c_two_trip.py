"""r2c of n = L1 x 2048 real points in two trips (decimated over the real data, FFTW_AMD_F_REAL_DEC) against the
three-trip plans FFTW_ESTIMATE picks (quarter lengths for powers of two, else half length) and the half-length
plan alone (FFTW_AMD_NO_RADIX4=1): ms per batch of 4 GiB of real input, whole % of the 8 TB/s
roofline on the algorithmic bytes (8 n in + 8 n out), error against torch.fft.rfft"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
buf = torch.rand((4 << 30) // 8, dtype=torch.float64, device="cuda") - 0.5
for n in ([int(v) for v in os.environ["PERF_N"].split(",")] if os.environ.get("PERF_N") else (1 << 22, 1 << 21, 1 << 20, 2048 * 1920, 2048 * 1000, 2048 * 1536, 2048 * 1280, 2048 * 1080, 2048 * 768, 2048 * 960, 2048 * 2000, 2048 * 1200)):
    hm = buf.numel() // n
    x = buf[:hm * n].reshape(hm, n)
    y = torch.zeros(hm, n // 2 + 1, dtype=torch.complex128, device="cuda")
    res = []
    for old in (0, 1, 2):
        if old: os.environ.pop("FFTW_AMD_REAL_DEC", None)
        else: os.environ["FFTW_AMD_REAL_DEC"] = "1"
        if old == 2: os.environ["FFTW_AMD_NO_RADIX4"] = "1"
        else: os.environ.pop("FFTW_AMD_NO_RADIX4", None)
        p = fa.plan_many_dft_r2c(1, [n], hm, x, None, 1, n, y, None, 1, n // 2 + 1)
        p.execute(); p.sync()
        ref = torch.fft.rfft(x[:2], dim=1)
        err = float((y[:2] - ref).abs().max() / ref.abs().max())
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        res.append((min(ts), err, len(p.steps())))
        del p
    os.environ.pop("FFTW_AMD_REAL_DEC", None)
    os.environ.pop("FFTW_AMD_NO_RADIX4", None)
    f = lambda t: 100 * 16.0 * n * hm / t / 8e12
    print("%8d x %-5d new %d trips %7.3f ms %5.1f %% err %.1e | old %d trips %7.3f ms %5.1f %% err %.1e | old, half length only: %d trips %7.3f ms %5.1f %%" %
          (n, hm, res[0][2], res[0][0] * 1e3, f(res[0][0]), res[0][1], res[1][2], res[1][0] * 1e3, f(res[1][0]), res[1][1], res[2][2], res[2][0] * 1e3, f(res[2][0])), flush=True)
    del y
