"""exploration / soak: random problems of every kind against the oracle for a fixed time
(not collected by pytest).  usage: python tests/stress_random.py SECONDS [SEED]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fftw3_amd as fa
from util import TOL, aerror, crand, rrand, oracle_dft, oracle_r2c, oracle_c2r, oracle_r2r

def rand_n(rng, budget):
    primes = [2, 2, 2, 2, 3, 3, 3, 5, 5, 7, 11, 13, 17, 19, 23]
    n = 1
    while True:
        f = primes[int(rng.integers(0, len(primes)))]
        if n * f > budget: break
        n *= f
        if rng.random() < 0.12: break
    if rng.random() < 0.08:
        n = int([17, 19, 23, 31, 37, 53, 97, 127, 257, 1031][int(rng.integers(0, 10))])
    return max(1, n)

def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t_end = time.time() + secs
    cases = fails = 0
    worst = 0.0
    while time.time() < t_end:
        kind = ["c2c", "r2c", "c2r", "r2r"][int(rng.integers(0, 4))]
        rank = int(rng.integers(1, 4))
        total_budget = 10 ** float(rng.uniform(2.0, 6.3))
        shape = [rand_n(rng, max(2.0, total_budget ** (1.0 / rank))) for _ in range(rank)]
        n = int(np.prod(shape))
        v = int(rng.integers(1, 6)) if n < 200000 else 1
        inplace = bool(rng.random() < 0.4)
        desc = (kind, shape, v, inplace)
        try:
            if kind == "c2c":
                sign = -1 if rng.random() < 0.5 else 1
                inter = bool(rng.random() < 0.3)
                x = crand(rng, v * n)
                xd = torch.from_numpy(x).cuda()
                yd = xd if inplace else torch.zeros_like(xd)
                st, di = (v, 1) if inter else (1, n)
                p = fa.plan_many_dft(rank, shape, v, xd, None, st, di, yd, None, st, di, sign)
                p.execute(); p.sync()
                want = np.zeros_like(x)
                oracle_dft(x, tuple(shape), v, sign, out=want, istride=st, idist=di, ostride=st, odist=di)
                e = aerror(yd.cpu().numpy(), want)
                desc += (sign, inter)
            elif kind == "r2c":
                hn = n // shape[-1] * (shape[-1] // 2 + 1)
                x = rrand(rng, v * n)
                xd = torch.from_numpy(x).cuda()
                yd = torch.zeros(v * hn, dtype=torch.complex128, device="cuda")
                p = fa.plan_many_dft_r2c(rank, shape, v, xd, None, 1, n, yd, None, 1, hn)
                p.execute(); p.sync()
                e = aerror(yd.cpu().numpy(), oracle_r2c(x, tuple(shape), v))
            elif kind == "c2r":
                hn = n // shape[-1] * (shape[-1] // 2 + 1)
                x = rrand(rng, v * n)
                y = oracle_r2c(x, tuple(shape), v)
                yd = torch.from_numpy(y).cuda()
                zd = torch.zeros(v * n, dtype=torch.float64, device="cuda")
                p = fa.plan_many_dft_c2r(rank, shape, v, yd, None, 1, hn, zd, None, 1, n)
                p.execute(); p.sync()
                e = aerror(zd.cpu().numpy(), oracle_c2r(y, tuple(shape), v))
            else:
                kinds = [int(rng.integers(0, 11)) for _ in range(rank)]
                shape = [max(2, s) if k == 3 else s for s, k in zip(shape, kinds)]
                n = int(np.prod(shape))
                if n > 300000:
                    continue
                x = rrand(rng, v * n)
                xd = torch.from_numpy(x).cuda()
                yd = xd if inplace else torch.zeros_like(xd)
                p = fa.plan_many_r2r(rank, shape, v, xd, None, 1, n, yd, None, 1, n, kinds)
                p.execute(); p.sync()
                e = aerror(yd.cpu().numpy(), oracle_r2r(x, shape, kinds, howmany=v))
                desc = (kind, shape, v, inplace, kinds)
        except Exception as ex:      # noqa: BLE001
            print("EXC", desc, repr(ex), flush=True)
            fails += 1
            continue
        cases += 1
        worst = max(worst, e)
        if not (e <= TOL):
            fails += 1
            print("FAIL", desc, e, p.sprint().replace("\n", " ")[:300], flush=True)
        if cases % 200 == 0:
            print("progress: %d cases, %d failures, worst %.2e" % (cases, fails, worst), flush=True)
    print("done: %d cases, %d failures, worst error %.3e" % (cases, fails, worst))
    return 1 if fails else 0

if __name__ == "__main__":
    sys.exit(main())
