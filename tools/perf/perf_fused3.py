import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20; b = int(os.environ.get("B", "2048"))
x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
res = {}
for tag, env in [("pipeline", {"FFTW_AMD_MIXED": "0"})] + [("mixed%d" % c, {"FFTW_AMD_MIXED": "1", "FFTW_AMD_MIXED_CHUNK": str(c)}) for c in (2, 3, 4, 6)]:
    os.environ.update(env)
    p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1)
    p.execute(); torch.cuda.synchronize()
    res[tag] = (p, [])
for rnd in range(5):
    for tag, (p, ts) in res.items():
        torch.cuda.synchronize(); t = time.perf_counter(); p.execute(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
for tag, (p, ts) in res.items():
    best = min(ts); med = sorted(ts)[len(ts) // 2]
    print("%-9s best %.2f us/xform (%.1f%%)  median %.2f" % (tag, best / b * 1e6, 32 * n * b / best / 8e12 * 100, med / b * 1e6), flush=True)
