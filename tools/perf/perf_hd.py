"""exploration (not a test): sizes whose strided axis is 1025 ... 2048 points long"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
def run(label, shape, hm):
    n = 1
    for v in shape: n *= v
    x = torch.view_as_complex(torch.rand((hm * n, 2), dtype=torch.float64, device="cuda") - 0.5)
    y = torch.zeros_like(x)
    for tuned in (1, 0):
        if tuned: os.environ.pop("FFTW_AMD_NO_NARROW", None)
        else: os.environ["FFTW_AMD_NO_NARROW"] = "1"
        p = fa.plan_many_dft(len(shape), list(shape), hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
        p.execute(); p.sync()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        import math
        print("%-22s %s %7.3f ms %6.0f GF %4.1f%%  %s" % (label, "narrow" if tuned else "before", t * 1e3, 5.0 * n * math.log2(n) * hm / t / 1e9,
              100 * 32.0 * n * hm / t / 8e12, " ".join(l.strip().split(" buf")[0] for l in p.sprint().splitlines()[1:])), flush=True)
        del p
run("2-D 1080 x 1920 x128", (1080, 1920), 128)
run("2-D 1200 x 1600 x128", (1200, 1600), 128)
run("1-D 2000*1000 x128", (2000000,), 128)
run("1-D 1080*1024 x256", (1080 * 1024,), 256)
run("3-D 128 x 1536 x 64 x32", (128, 1536, 64), 32)
