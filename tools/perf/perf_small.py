"""exploration: short transforms in large batches, one-stage rows kernel against what ran before (not a test)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
SIZES = list(range(2, 33)) + [36, 45, 64, 100, 128, 256, 512, 1024]
for n in SIZES:
    hm = (1 << 27) // n
    x = torch.view_as_complex(torch.rand((hm * n, 2), dtype=torch.float64, device="cuda") - 0.5)
    y = torch.zeros_like(x)
    for no_r1 in (("1", "") if n <= 32 else ("",)):
        if no_r1: os.environ["FFTW_AMD_NO_R1"] = "1"
        else: os.environ.pop("FFTW_AMD_NO_R1", None)
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
        for _ in range(2): p.execute()
        p.sync()
        t0 = time.perf_counter()
        for _ in range(5): p.execute()
        p.sync()
        dt = (time.perf_counter() - t0) / 5
        gb = 32.0 * n * hm / 1e9
        print("n=%-5d x%-8d %7.3f ms  %6.0f GB/s (%4.1f%%)  %s" % (n, hm, dt * 1e3, gb / dt, gb / dt / 80,
              " ".join(l.strip().split(" tile")[0].lstrip("(") for l in p.sprint().splitlines()[1:])), flush=True)
        del p
