"""exploration (not a test): every two-pass split L1 x L2 of a few mixed-radix lengths whose factors both have a
register kernel, against the planner's default"""
import os, sys, time, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
x = torch.view_as_complex(torch.rand(((4 << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
menu = [L for L in range(16, 1025) if fa.lib.fa_hip_rr_tile(L) > 0 or fa.lib.fa_hip_r3t_tile(L) > 0 or L == 1024]
out = open("gpurun_out/splits2.jsonl", "a")
def run(n, forced):
    hm = x.numel() // n
    if forced: os.environ["FFTW_AMD_FORCE_LENS"] = ",".join(str(v) for v in forced)
    else: os.environ.pop("FFTW_AMD_FORCE_LENS", None)
    p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    p.execute(); p.sync()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
    lens = [int(l.strip().split("pass-")[1].split("/")[0]) for l in p.sprint().splitlines()[1:]]
    prof = p.execute_profiled()
    return min(ts) * 1e3, lens, [m for _, m, l in prof], hm
for n in [int(v) for v in os.environ.get("NS", "1000000,60060,100000,10000,65536,46656,250000,518400,93312,200000,123200").split(",")]:
    td, ld, _, hm = run(n, None)
    res = []
    for a in menu:
        if n % a == 0 and (n // a) in menu and max(a, n // a) <= 8 * min(a, n // a):
            t, l, st, hm = run(n, (a, n // a))
            res.append((t, (a, n // a)))
            out.write(json.dumps({"n": n, "b": hm, "lens": [a, n // a], "ms": t, "steps": st, "gib_per_pass": 32.0 * n * hm / 2**30}) + "\n"); out.flush()
    res.sort()
    print("%8d default %s %.3f ms | best %s %.3f ms (%+.1f %%) | %s" % (n, ld, td, res[0][1] if res else None, res[0][0] if res else 0,
          100 * (td / res[0][0] - 1) if res else 0, " ".join("%dx%d:%.2f" % (a, b, t) for t, (a, b) in res[:6])), flush=True)
