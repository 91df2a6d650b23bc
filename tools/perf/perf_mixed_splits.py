"""exploration (not a test): three-pass splits of N = 3*5*7*11*13*2^10 (BASELINE configs[3]) whose
factors all have a register kernel, through the FFTW_AMD_FORCE_LENS hook.  One process per
candidate is not needed: the hook is read at plan time."""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
import json
n = int(os.environ.get("N", str(3 * 5 * 7 * 11 * 13 * 1024)))
b = max(2, int((int(os.environ.get("GIB", "8")) << 30) // (16 * n)))
x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
out = open(os.environ.get("OUT", "gpurun_out/splits.jsonl"), "a")
menu = [L for L in range(16, 1025) if fa.lib.fa_hip_rr_tile(L) > 0 or fa.lib.fa_hip_r3t_tile(L) > 0 or L == 1024]
cands = []
for a in menu:
    if n % a: continue
    for c in menu:
        if (n // a) % c: continue
        m = n // a // c
        if m in menu and max(a, c, m) <= 4 * min(a, c, m):
            cands.append((a, m, c))
import random
random.Random(n).shuffle(cands)
cands = cands[:int(os.environ.get("MAXC", "120"))]
print(n, b, len(cands), "candidate splits", flush=True)
res = []
for lens in cands:
    os.environ["FFTW_AMD_FORCE_LENS"] = ",".join(str(v) for v in lens)
    try:
        p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1, fa.ESTIMATE)
    except Exception as e:
        continue
    for _ in range(1): p.execute()
    p.sync()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
    prof = p.execute_profiled()
    t = min(ts)
    res.append((t, lens, [round(m, 2) for _, m, l in prof]))
    out.write(json.dumps({"n": n, "b": b, "lens": lens, "ms": t * 1e3, "steps": [m for _, m, l in prof], "gib_per_pass": 32.0 * n * b / 2**30}) + "\n")
    out.flush()
    print("%s: %.3f ms per %d transforms  steps %s" % (lens, t * 1e3, b, res[-1][2]), flush=True)
    del p
res.sort()
print("best:")
for t, lens, st in res[:12]:
    print("  %s %.3f ms %s  -> %.0f GFLOPS" % (lens, t * 1e3, st, 5.0 * n * __import__("math").log2(n) * b / t / 1e9))
