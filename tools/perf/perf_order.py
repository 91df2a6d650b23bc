"""exploration: pass order of two-pass splits (not a test)"""
import os, time, torch
import fftw3_amd as fa
def run(label, mk):
    p = mk()
    for _ in range(2): p.execute()
    p.sync()
    t0 = time.perf_counter()
    for _ in range(5): p.execute()
    p.sync()
    dt = (time.perf_counter() - t0) / 5
    prof = p.execute_profiled()
    print("%-34s %8.3f ms  steps(ms/launch): %s" % (label, dt * 1e3, " ".join("%.3f" % (ms / c) for _, ms, c in prof)), flush=True)
    print("    " + p.sprint().replace("\n", " "))
for lf in (0, 1):
    os.environ["FFTW_AMD_LONG_FIRST"] = str(lf)
    for lg in (19, 18, 17, 16):
        n, hm = 1 << lg, (1 << 29) >> lg
        x = torch.view_as_complex(torch.rand((hm * n, 2), dtype=torch.float64, device="cuda") - 0.5)
        y = torch.zeros_like(x)
        run("c2c 2^%d x%d long_first=%d" % (lg, hm, lf), lambda: fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD))
        del x, y
    n, hm = 1 << 20, 512
    xr = torch.rand(hm * n, dtype=torch.float64, device="cuda") - 0.5
    yr = torch.zeros_like(xr)
    run("REDFT10 2^20 x512 long_first=%d" % lf, lambda: fa.plan_many_r2r(1, [n], hm, xr, None, 1, n, yr, None, 1, n, [fa.REDFT10]))
    z = torch.zeros(hm * (n // 2 + 1), dtype=torch.complex128, device="cuda")
    run("r2c 2^20 x512 long_first=%d" % lf, lambda: fa.plan_many_dft_r2c(1, [n], hm, xr, None, 1, n, z, None, 1, n // 2 + 1))
    del xr, yr, z
