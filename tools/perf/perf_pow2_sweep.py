"""exploration (not a test): batched c2c powers of two 2^10 ... 2^24, 8 GiB of input each, against the scratch
chunk size: does the Infinity-Cache policy that pays at 2^20 pay elsewhere?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
gib = int(os.environ.get("GIB", "8"))
x = torch.view_as_complex(torch.rand(((gib << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
for lg in [int(v) for v in os.environ.get("LGS", "10,12,14,16,17,18,19,20,21,22,24").split(",")]:
    n = 1 << lg
    hm = x.numel() // n
    line = "2^%-2d x%-8d" % (lg, hm)
    for mib in [int(v) for v in os.environ.get("CHUNKS", "128,256,512,1024").split(",")]:
        fa.set_chunk_bytes(mib << 20)
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
        p.execute(); p.sync()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        line += " | %4d MiB: %6.2f ms %5.0f GF %4.1f%%" % (mib, t * 1e3, 5.0 * n * lg * hm / t / 1e9, 100 * 32.0 * n * hm / t / 8e12)
        if mib == 256:
            plan = " ".join(l.strip().split(" tile")[0] for l in p.sprint().splitlines()[1:])
        del p
    print(line + "   " + plan, flush=True)
