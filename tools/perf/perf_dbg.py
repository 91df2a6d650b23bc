import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20
fa.set_chunk_bytes(1 << 40)
for b in (4, 64):
    x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
    p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1)
    for it in range(3): p.execute()
    torch.cuda.synchronize()
    prof = p.execute_profiled(); prof = p.execute_profiled()
    print("DBG=%s batch %3d: steps us/xform=%s" % (os.environ.get("FFTW_AMD_DBG"), b, [round(t[1] * 1e3 / b, 2) for t in prof]), flush=True)
