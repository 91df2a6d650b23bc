"""one execution of c2c 2^19 (pass-512 + pass-1024) and c2c 2^20 at reduced batch, for rocprofv3 --pmc passes"""
import torch
import fftw3_amd as fa
for lg in (19, 20):
    n, hm = 1 << lg, 256
    x = torch.view_as_complex(torch.rand((hm * n, 2), dtype=torch.float64, device="cuda") - 0.5)
    y = torch.zeros_like(x)
    p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    for _ in range(3):
        p.execute()
    p.sync()
