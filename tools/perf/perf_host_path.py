"""exploration (not a test): throughput an unmodified host caller sees (pinned host arrays, staged through PCIe),
with and without the chunked host pipeline"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
n, b = 1 << 20, 128
x = torch.empty((b, n), dtype=torch.complex128).pin_memory()
x.real.uniform_(-0.5, 0.5); x.imag.uniform_(-0.5, 0.5)
y = torch.empty((b, n), dtype=torch.complex128).pin_memory()
p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, fa.FORWARD)
for _ in range(2): p.execute()
ts = []
for _ in range(4):
    t0 = time.perf_counter(); p.execute(); ts.append(time.perf_counter() - t0)
t = min(ts)
print("FFTW_AMD_HOST_PIPELINE=%s: %d x 2^20 on pinned host arrays: %.1f ms, %.0f GFLOPS, %.1f GB/s each way" % (
    os.environ.get("FFTW_AMD_HOST_PIPELINE", "1"), b, t * 1e3, 5.0 * n * 20 * b / t / 1e9, 16.0 * n * b / t / 1e9))
