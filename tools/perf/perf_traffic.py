"""one execution of the headline plan at reduced batch, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20
b = int(os.environ.get("B", "512"))
x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1)
print(p.sprint())
for it in range(3): p.execute()
torch.cuda.synchronize()
