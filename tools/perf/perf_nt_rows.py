"""exploration (not a test): dense rows, nontemporal accesses off (FFTW_AMD_NT=0) against forced on (=2); 2 GiB per array"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
tot = (2 << 30) // 16
x = (torch.rand(tot, dtype=torch.float64, device="cuda") - 0.5).to(torch.complex128)
y = torch.zeros_like(x)
SIZES = [int(a) for a in sys.argv[1:]] or [33, 35, 36, 40, 45, 48, 49, 50, 60, 64, 72, 80, 96, 100, 120, 125, 143, 144, 200, 240, 243, 256, 300, 343, 400, 500, 512, 600,
         625, 640, 700, 729, 1000, 1001, 1024, 1100, 1536, 2000, 2048, 3000, 3003, 4000, 4096, 5000, 6561, 7680, 8192]
for n in SIZES:
    hm = tot // n
    row = []
    for nt in ("0", "2"):
        os.environ["FFTW_AMD_NT"] = nt
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD, fa.ESTIMATE)
        p.execute(); p.sync()
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        row.append((t, " ".join(l.strip().split(" tile")[0].lstrip("(") for l in p.sprint().splitlines()[1:])))
        del p
    print("n=%-5d nt=0 %7.3f ms %4.1f%%   nt=2 %7.3f ms %4.1f%%   %+5.1f%%  [%s]" % (n, row[0][0] * 1e3, 100 * 32.0 * n * hm / row[0][0] / 8e12,
          row[1][0] * 1e3, 100 * 32.0 * n * hm / row[1][0] / 8e12, 100 * (row[0][0] / row[1][0] - 1), row[1][1]), flush=True)
