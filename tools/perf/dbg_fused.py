import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20
b = int(os.environ.get("B", "8"))
x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.zeros_like(x)
p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1)
print("plan ok", flush=True)
t = time.perf_counter()
p.execute(); torch.cuda.synchronize()
print("executed in %.3f s" % (time.perf_counter() - t), flush=True)
ref = torch.fft.fft(x, dim=1)
print("err", float((y - ref).abs().max() / ref.abs().max()), flush=True)
