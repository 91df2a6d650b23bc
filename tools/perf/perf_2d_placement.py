"""exploration (not a test): does the 2-D 4096^2 x 64 plan care where its arrays lie?  x and y are views into larger
allocations at different byte offsets, with and without other allocations made (and freed) first"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
n, hm = 4096 * 4096, 64
def run(tag, xoff, yoff, junk_gib=0):
    junk = [torch.empty((g << 30), dtype=torch.uint8, device="cuda") for g in ([junk_gib] if junk_gib else [])]
    del junk
    torch.cuda.empty_cache()
    xb = torch.zeros(n * hm + (1 << 22), dtype=torch.complex128, device="cuda")
    yb = torch.zeros(n * hm + (1 << 22), dtype=torch.complex128, device="cuda")
    x = xb[xoff // 16: xoff // 16 + n * hm]; y = yb[yoff // 16: yoff // 16 + n * hm]
    p = fa.plan_many_dft(2, [4096, 4096], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    for _ in range(2): p.execute()
    p.sync()
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
    prof = p.execute_profiled()
    print("%-28s x %#x y %#x  %7.3f ms  steps %s" % (tag, x.data_ptr(), y.data_ptr(), min(ts) * 1e3, [round(m, 2) for _, m, l in prof]), flush=True)
    del p, x, y, xb, yb
    torch.cuda.empty_cache()
run("fresh", 0, 0)
run("y + 4 KiB", 0, 4096)
run("y + 64 KiB + 128", 0, 65536 + 128)
run("y + 1 MiB + 4 KiB", 0, (1 << 20) + 4096)
run("x + 4 KiB", 4096, 0)
run("after 128 GiB junk", 0, 0, 128)
run("after 64 GiB junk", 0, 0, 64)
run("fresh again", 0, 0)
