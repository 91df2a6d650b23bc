"""exploration: the planner's three-pass pick against the best split measured so far for the sweep lengths"""
import os, sys, time, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
rows = [json.loads(l) for l in open("profiles/r02_three_pass_split_samples.jsonl")]
x = torch.view_as_complex(torch.rand(((8 << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
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
    return min(ts) * 1e3, lens
for n in sorted(set(r["n"] for r in rows)):
    best = min((r for r in rows if r["n"] == n), key=lambda r: r["ms"])
    tb, lb = run(n, best["lens"])
    tp, lp = run(n, None)
    print("%9d  planner %s %.2f ms | best sampled %s %.2f ms | %+.1f %%" % (n, lp, tp, lb, tb, 100 * (tp / tb - 1)), flush=True)
