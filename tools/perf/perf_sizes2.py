"""exploration: sizes that use the strided three-stage kernels (not a test)"""
import time, torch
import fftw3_amd as fa
def run(label, shape, hm, stride=None):
    n = 1
    for v in shape: n *= v
    x = torch.view_as_complex(torch.rand((hm * n, 2), dtype=torch.float64, device="cuda") - 0.5)
    y = torch.zeros_like(x)
    if stride:
        p = fa.plan_many_dft(len(shape), list(shape), hm, x, None, hm, 1, y, None, hm, 1, fa.FORWARD)
    else:
        p = fa.plan_many_dft(len(shape), list(shape), hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    for _ in range(2): p.execute()
    p.sync()
    t0 = time.perf_counter()
    for _ in range(5): p.execute()
    p.sync()
    dt = (time.perf_counter() - t0) / 5
    gb = 32.0 * n * hm / 1e9
    print("%-26s %7.3f ms  %6.0f GB/s alg (%4.1f%%)  %s" % (label, dt * 1e3, gb / dt, gb / dt / 80, p.sprint().replace("\n", " ")[:170]), flush=True)
run("1d 10^6 x64", (1000000,), 64)
run("1d 720^2 x128", (518400,), 128)
run("1d 1000 interleaved x65536", (1000,), 65536, stride=True)
run("2d 1000x1000 x64", (1000, 1000), 64)
run("3d 200^3 x8", (200, 200, 200), 8)
run("3d 100^3 x64", (100, 100, 100), 64)
run("2d 1920x1080 x32", (1080, 1920), 32)
