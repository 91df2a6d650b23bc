"""perf exploration (not a test): cfg2 unit (c2c 2^20, out of place) against the scratch
chunk size and the nontemporal policy.  FFTW_AMD_NT is read once per process, so the
policy comes from the environment: run once per policy."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20
b = int(os.environ.get("B", "1024"))
x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
print("FFTW_AMD_NT=%s batch %d" % (os.environ.get("FFTW_AMD_NT", "(default)"), b), flush=True)
for mib in [int(v) for v in os.environ.get("CHUNKS", "32,64,96,128,192,256,384,512,1024").split(",")]:
    fa.set_chunk_bytes(mib << 20)
    p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1, fa.ESTIMATE)
    for _ in range(2): p.execute()
    p.sync()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
    prof = p.execute_profiled()
    t = min(ts)
    print("chunk %5d MiB (%3d xforms): %.3f ms  = %.2f us/xform  whole %.1f%%   steps(ms/launch) %s" % (
        mib, p.chunk, t * 1e3, t / b * 1e6, 100 * 32.0 * n * b / t / 8e12,
        " ".join("%.4f" % (m / max(1, l)) for _, m, l in prof)), flush=True)
    del p
