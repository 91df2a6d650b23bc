"""perf exploration: one configuration, few iterations (for rocprofv3 counter passes)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20
b = int(os.environ.get("B", "64"))
fa.set_chunk_bytes(1 << 40)
x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1)
for it in range(int(os.environ.get("REPS", "5"))): p.execute()
torch.cuda.synchronize()
