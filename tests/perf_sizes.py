"""perf exploration: a sweep of sizes (power-of-two and mixed radix), batch ~1 GiB per array"""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
sizes = [int(v) for v in os.environ.get("SIZES", "1000,5000,15015,60060,59049,78125,1000000,1024,16384,65536,262144,1048576,4194304").split(",")]
for n in sizes:
    b = max(1, (1 << 30) // (16 * n))
    x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
    p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1)
    p.execute(); torch.cuda.synchronize()
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); p.execute(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print("n=%8d b=%7d: %8.3f ms  %6.0f GFLOPS  %5.1f%% of roofline  %s" % (n, b, best * 1e3, 5 * n * math.log2(n) * b / best / 1e9,
          32 * n * b / best / 8e12 * 100, " ".join(l.strip() for l in p.sprint().split("\n")[1:])[:110]), flush=True)
    del x, y, p
