"""perf exploration: load-only rate when the input cycles through P distinct transforms (P*16 MiB working set)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20
fa.set_chunk_bytes(1 << 40)
for P in (1, 2, 4, 8, 12, 16, 32):
    R = 256 // P
    x = torch.randn(P, n, dtype=torch.complex128, device=dev); y = torch.empty(R * P, n, dtype=torch.complex128, device=dev)
    mode = os.environ.get("MODE", "in")
    if mode == "in":
        p = fa.plan_guru64_dft([(n, 1, 1)], [(P, n, n), (R, 0, P * n)], x, y, -1)
    else:  # alias the output instead: input distinct
        xin = torch.randn(R * P, n, dtype=torch.complex128, device=dev)
        p = fa.plan_guru64_dft([(n, 1, 1)], [(P, n, n), (R, P * n, 0)], xin, y, -1)
    for it in range(2): p.execute()
    torch.cuda.synchronize()
    prof = p.execute_profiled(); prof = p.execute_profiled()
    print("DBG=%s mode=%s P=%d (%d MiB): steps us/xform=%s  %s" % (os.environ.get("FFTW_AMD_DBG"), mode, P, P * 16,
          [round(t[1] * 1e3 / (R * P), 2) for t in prof], p.sprint().split("\n")[1]), flush=True)
    del x, y
