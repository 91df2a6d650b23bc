"""perf exploration (not a test): chunk-size sweep for N=2^20"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20
b = int(os.environ.get("B", "512"))
x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
for mib in [int(v) for v in os.environ.get("CHUNKS", "16,32,64,128,256,1024").split(",")]:
    fa.set_chunk_bytes(mib << 20)
    p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1)
    p.execute(); torch.cuda.synchronize()
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize(); t = time.perf_counter(); p.execute(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    prof = p.execute_profiled()
    print("chunk %5d MiB (=%d xforms): %.3f ms  %.0f GFLOPS  %.0f GB/s alg (%.1f%%)  steps(ms)=%s launches=%d" % (
        mib, p.chunk, best * 1e3, 5 * n * 20 * b / best / 1e9, 32 * n * b / best / 1e9, 32 * n * b / best / 8e12 * 100,
        [round(t[1], 3) for t in prof], prof[0][2]), flush=True)
