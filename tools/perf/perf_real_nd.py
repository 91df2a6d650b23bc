"""exploration: multi-dimensional real transforms with / without the fused rows kernel (not a test)"""
import os, sys, time, torch
import fftw3_amd as fa
def run(label, shape, hm):
    n = 1
    for v in shape: n *= v
    hs = n // shape[-1] * (shape[-1] // 2 + 1)
    x = torch.rand(hm * n, dtype=torch.float64, device="cuda") - 0.5
    y = torch.zeros(hm * hs, dtype=torch.complex128, device="cuda")
    p = fa.plan_many_dft_r2c(len(shape), list(shape), hm, x, None, 1, n, y, None, 1, hs)
    for _ in range(2): p.execute()
    p.sync()
    t0 = time.perf_counter()
    for _ in range(5): p.execute()
    p.sync()
    dt = (time.perf_counter() - t0) / 5
    gb = (8.0 * n + 16.0 * hs) * hm / 1e9
    print("%-22s %7.3f ms  %6.0f GB/s alg  %s" % (label, dt * 1e3, gb / dt, p.sprint().replace("\n", " ")[:200]), flush=True)
print("NO_R2CROWS =", os.environ.get("FFTW_AMD_NO_R2CROWS"))
run("1d 1024 x262144", (1024,), 262144)
run("1d 2048 x131072", (2048,), 131072)
run("2d 1024x1024 x128", (1024, 1024), 128)
run("2d 4096x2048 x16", (4096, 2048), 16)
run("3d 256^3 x8", (256, 256, 256), 8)
run("3d 512^3 x1", (512, 512, 512), 1)
