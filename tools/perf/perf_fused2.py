import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20; b = 512
x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1)
p.execute(); torch.cuda.synchronize()
best = 1e9
for it in range(4):
    torch.cuda.synchronize(); t = time.perf_counter(); p.execute(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
e = float((y[:2] - torch.fft.fft(x[:2], dim=1)).abs().max() / 1024)
print("DBG=%s LAG=%s MIXED_CHUNK=%s: %.2f us/xform (%.1f%%)  err %.1e" % (os.environ.get("FFTW_AMD_FUSED_DBG"), os.environ.get("FFTW_AMD_FUSED_LAG"), os.environ.get("FFTW_AMD_MIXED_CHUNK"), best / b * 1e6, 32 * n * b / best / 8e12 * 100, e), flush=True)
