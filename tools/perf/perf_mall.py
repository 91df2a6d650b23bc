"""perf exploration: does the Infinity Cache help when everything is resident?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20
fa.set_chunk_bytes(1 << 40)
for b in (1, 2, 4, 8, 16, 64):
    x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
    p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1)
    for it in range(3): p.execute()
    torch.cuda.synchronize(); t = time.perf_counter()
    reps = 50
    for it in range(reps): p.execute()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    prof = p.execute_profiled()
    print("batch %3d (%4d MiB in+out+scratch): %.1f us/xform  %.0f GB/s alg  steps us/xform=%s" % (
        b, b * 48, dt / b * 1e6, 32 * n * b / dt / 1e9, [round(t[1] * 1e3 / b, 2) for t in prof]), flush=True)
