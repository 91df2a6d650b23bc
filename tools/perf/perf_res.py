"""perf exploration: load-only / store-only rates when every batch element aliases the same 16 MiB"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20
b = 256
fa.set_chunk_bytes(1 << 40)
mode = os.environ.get("MODE", "alias_in")
x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
idist = 0 if "in" in mode else n
odist = 0 if "out" in mode else n
p = fa.plan_many_dft(1, [n], b, x, None, 1, idist, y, None, 1, odist, -1)
for it in range(2): p.execute()
torch.cuda.synchronize()
prof = p.execute_profiled(); prof = p.execute_profiled()
print("DBG=%s mode=%s: steps us/xform=%s" % (os.environ.get("FFTW_AMD_DBG"), mode, [round(t[1] * 1e3 / b, 2) for t in prof]), flush=True)
