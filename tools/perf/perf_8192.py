"""exploration (not a test): rows of 8192 in one trip (pass3s<32>) against the two-pass split (FFTW_AMD_NO_3S=1);
8 GiB of complex128, out of place and in place"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
hm = (8 << 30) // 16 // n
x = (torch.rand(hm * n, dtype=torch.float64, device="cuda") - 0.5).to(torch.complex128)
y = torch.zeros_like(x)
for no3s in (1, 0, 1, 0):
    if no3s: os.environ["FFTW_AMD_NO_3S"] = "1"
    else: os.environ.pop("FFTW_AMD_NO_3S", None)
    for dst in (y, x):
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, dst, None, 1, n, fa.FORWARD, fa.ESTIMATE)
        p.execute(); p.sync()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        print("%s %-8s %7.3f ms %6.0f GFLOPS whole %4.1f%%  %s" % ("two-pass" if no3s else "one-trip", "in place" if dst is x else "",
              t * 1e3, 5.0 * n * __import__("math").log2(n) * hm / t / 1e9, 100 * 32.0 * n * hm / t / 8e12,
              " ".join(l.strip() for l in p.sprint().splitlines()[1:])), flush=True)
        del p
