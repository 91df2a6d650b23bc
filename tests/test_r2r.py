"""r2r family (R2HC, HC2R, DHT, REDFT00/01/10/11, RODFT00/01/10/11): SURVEY.md
section 8(f) row 3.

CPU tier: the oracle's r2r restatement is pinned by the reference's own r2r
verifier (tests/verifier.py <- fftw/libbench2/verify-r2r.c), by the defining
sums of the reference manual (oracle_r2r_direct) and by scipy's independent
DCT/DST; the planner's step lists are executed by the numpy step interpreter.
GPU tier: the HIP path through the C-ABI against the oracle, plus
size-independent inverse-pair properties at sizes the oracle cannot reach.
"""
import numpy as np
import pytest

import fftw3_amd as fa
import verifier as V
from step_interp import run_plan_on_host
from util import TOL, aerror, oracle_r2r, oracle_r2r_direct, rrand

KINDS = list(range(11))
NAMES = ["R2HC", "HC2R", "DHT", "REDFT00", "REDFT01", "REDFT10", "REDFT11",
         "RODFT00", "RODFT01", "RODFT10", "RODFT11"]
# unnormalised inverse of every kind and the factor kind^-1(kind(x)) / x
# (reference manual, fftw/doc/reference.texi:2135-2150, :2232-2244)
INVERSE = {fa.R2HC: fa.HC2R, fa.HC2R: fa.R2HC, fa.DHT: fa.DHT,
           fa.REDFT00: fa.REDFT00, fa.REDFT01: fa.REDFT10, fa.REDFT10: fa.REDFT01,
           fa.REDFT11: fa.REDFT11, fa.RODFT00: fa.RODFT00, fa.RODFT01: fa.RODFT10,
           fa.RODFT10: fa.RODFT01, fa.RODFT11: fa.RODFT11}


def logical_n(kind, n):
    return V._logical_n(kind, n)


def sizes_for(kind, sizes):
    return [n for n in sizes if not (kind == fa.REDFT00 and n < 2)]


# --------------------------------------------------------------- oracle pin

def _scipy_ref(x, k):
    import scipy.fft as sf
    n = x.size
    if k == fa.R2HC:
        F = np.fft.rfft(x)
        y = np.zeros(n)
        y[:n // 2 + 1] = F.real
        for j in range(1, (n + 1) // 2):
            y[n - j] = F[j].imag
        return y
    if k == fa.HC2R:
        F = np.zeros(n // 2 + 1, complex)
        F.real = x[:n // 2 + 1]
        for j in range(1, (n + 1) // 2):
            F[j] += 1j * x[n - j]
        return np.fft.irfft(F, n) * n
    if k == fa.DHT:
        F = np.fft.fft(x)
        return F.real - F.imag
    fam, typ = {fa.REDFT00: ("c", 1), fa.REDFT01: ("c", 3), fa.REDFT10: ("c", 2), fa.REDFT11: ("c", 4),
                fa.RODFT00: ("s", 1), fa.RODFT01: ("s", 3), fa.RODFT10: ("s", 2),
                fa.RODFT11: ("s", 4)}[k]
    return (sf.dct if fam == "c" else sf.dst)(x, type=typ)


@pytest.mark.parametrize("kind", KINDS, ids=NAMES)
def test_oracle_r2r_matches_definitions(kind):
    """oracle_r2r_many (symmetric extension + pinned r2c) == defining sums == scipy"""
    rng = np.random.default_rng(kind)
    for n in sizes_for(kind, [1, 2, 3, 4, 5, 7, 8, 9, 16, 31, 64, 100, 127, 360]):
        x = rrand(rng, n)
        a = oracle_r2r(x, [n], [kind])
        assert aerror(a, oracle_r2r_direct(x, kind)) <= TOL, (NAMES[kind], n)
        assert aerror(a, _scipy_ref(x, kind)) <= TOL, (NAMES[kind], n)


@pytest.mark.parametrize("kind", KINDS, ids=NAMES)
def test_oracle_r2r_reference_verifier(kind):
    """the reference's r2r self-test (impulse response, linearity, time shift)"""
    for n in sizes_for(kind, [1, 2, 3, 4, 6, 9, 16, 25, 60]):
        def apply(x, n=n):
            v = x.shape[0]
            return oracle_r2r(x.reshape(-1), [n], [kind], howmany=v).reshape(v, n)
        V.verify_r2r(apply, (n,), [kind], vecn=2, seed=n, rounds=3)


def test_oracle_r2r_known_answers():
    """closed forms: constant input and single cosines"""
    n = 16
    one = np.ones(n)
    y = oracle_r2r(one, [n], [fa.REDFT10])
    want = np.zeros(n)
    want[0] = 2 * n
    assert aerror(y + 1.0, want + 1.0) <= TOL
    # DCT-II of cos(pi (j+1/2) m / n) is n at k = m
    m = 5
    x = np.cos(np.pi * (np.arange(n) + 0.5) * m / n)
    want = np.zeros(n)
    want[m] = n
    assert aerror(oracle_r2r(x, [n], [fa.REDFT10]) + 1.0, want + 1.0) <= TOL
    # DST-I of sin(pi (j+1)(m+1)/(n+1)) is (n+1) at k = m
    x = np.sin(np.pi * (np.arange(n) + 1) * (m + 1) / (n + 1))
    want = np.zeros(n)
    want[m] = n + 1
    assert aerror(oracle_r2r(x, [n], [fa.RODFT00]) + 1.0, want + 1.0) <= TOL
    # R2HC of a real cosine
    x = np.cos(2 * np.pi * 3 * np.arange(n) / n)
    want = np.zeros(n)
    want[3] = n / 2
    assert aerror(oracle_r2r(x, [n], [fa.R2HC]) + 1.0, want + 1.0) <= TOL


def test_oracle_r2r_multidim_separable():
    rng = np.random.default_rng(3)
    shape, kinds = [5, 6, 4], [fa.REDFT10, fa.RODFT00, fa.DHT]
    x = rrand(rng, *shape)
    a = oracle_r2r(x.reshape(-1), shape, kinds).reshape(shape)
    b = x.copy()
    for d, k in enumerate(kinds):
        b = np.apply_along_axis(lambda v: oracle_r2r_direct(v, k), d, b)
    assert aerror(a, b) <= TOL


# --------------------------------------------------------- planner (CPU tier)

SIZES = [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 31, 64, 100, 127, 128, 1000, 4096, 10007]


@pytest.mark.parametrize("kind", KINDS, ids=NAMES)
def test_planner_r2r_1d(kind):
    rng = np.random.default_rng(100 + kind)
    for n in sizes_for(kind, SIZES):
        x = rrand(rng, n)
        y = np.zeros(n)
        p = fa.plan_r2r_1d(n, x, y, kind)
        x0 = x.copy()
        run_plan_on_host(p, x, y)
        assert aerror(y, oracle_r2r(x0, [n], [kind])) <= TOL, (NAMES[kind], n)
        assert np.array_equal(x, x0)                     # out of place: input preserved


@pytest.mark.parametrize("shape,kinds,howmany", [
    ([8, 12], [fa.REDFT10, fa.REDFT00], 3),
    ([7, 9], [fa.REDFT11, fa.RODFT11], 2),
    ([4, 5, 6], [fa.R2HC, fa.DHT, fa.RODFT01], 2),
    ([16, 16], [fa.REDFT10, fa.REDFT10], 1),
    ([3, 1, 5], [fa.REDFT01, fa.RODFT10, fa.HC2R], 4),
    ([33, 20], [fa.RODFT00, fa.REDFT00], 2),
    ([64, 64], [fa.REDFT01, fa.REDFT01], 1),
])
def test_planner_r2r_multidim(shape, kinds, howmany):
    rng = np.random.default_rng(7)
    n = int(np.prod(shape))
    x = rrand(rng, howmany * n)
    y = np.zeros(howmany * n)
    p = fa.plan_many_r2r(len(shape), shape, howmany, x, None, 1, n, y, None, 1, n, kinds)
    run_plan_on_host(p, x, y)
    assert aerror(y, oracle_r2r(x, shape, kinds, howmany=howmany)) <= TOL


def test_planner_r2r_strided_embedded():
    rng = np.random.default_rng(8)
    shape, kinds, hm = [6, 10], [fa.REDFT10, fa.RODFT01], 3
    ie, oe, istr, ostr = [7, 12], [6, 11], 2, 3
    idist = ie[0] * ie[1] * istr + 5
    odist = oe[0] * oe[1] * ostr + 1
    x = rrand(rng, hm * idist)
    y = np.full(hm * odist, 7.0)
    p = fa.plan_many_r2r(2, shape, hm, x, ie, istr, idist, y, oe, ostr, odist, kinds)
    run_plan_on_host(p, x, y)
    want = np.full(hm * odist, 7.0)
    oracle_r2r(x, shape, kinds, howmany=hm, out=want, inembed=ie, istride=istr, idist=idist,
               onembed=oe, ostride=ostr, odist=odist)
    assert aerror(y, want) <= TOL
    assert np.array_equal(y == 7.0, want == 7.0)         # gaps untouched


@pytest.mark.parametrize("kind", KINDS, ids=NAMES)
def test_planner_r2r_inplace_and_chunked(kind):
    rng = np.random.default_rng(9)
    n, hm = 24, 37
    x = rrand(rng, hm * n)
    x0 = x.copy()
    fa.set_chunk_bytes(4096)
    try:
        p = fa.plan_many_r2r(1, [n], hm, x, None, 1, n, x, None, 1, n, [kind])
        assert p.chunk < p.batch
    finally:
        fa.set_chunk_bytes(0)
    run_plan_on_host(p, x, x)
    assert aerror(x, oracle_r2r(x0, [n], [kind], howmany=hm)) <= TOL


def test_planner_r2r_transposed_batch():
    rng = np.random.default_rng(10)
    n, hm = 20, 6
    x = rrand(rng, hm * n)
    y = np.zeros(hm * n)
    p = fa.plan_many_r2r(1, [n], hm, x, None, hm, 1, y, None, hm, 1, [fa.REDFT11])
    run_plan_on_host(p, x, y)
    want = np.zeros(hm * n)
    oracle_r2r(x, [n], [fa.REDFT11], howmany=hm, out=want, istride=hm, idist=1, ostride=hm, odist=1)
    assert aerror(y, want) <= TOL


def test_planner_r2r_guru_and_rank0():
    rng = np.random.default_rng(11)
    n0, n1, hm = 6, 9, 2
    x = rrand(rng, hm * n0 * n1)
    y = np.zeros(hm * n0 * n1)
    # column-major view of the same data: dims (n1 stride 1? no) -- plain row-major through guru
    p = fa.plan_guru64_r2r([(n0, n1, n1), (n1, 1, 1)], [(hm, n0 * n1, n0 * n1)], x, y,
                           [fa.RODFT10, fa.REDFT01])
    run_plan_on_host(p, x, y)
    assert aerror(y, oracle_r2r(x, [n0, n1], [fa.RODFT10, fa.REDFT01], howmany=hm)) <= TOL
    # rank 0 = copy of the howmany loop
    src = rrand(rng, 10)
    dst = np.zeros(20)
    p = fa.plan_guru64_r2r([], [(10, 1, 2)], src, dst, [])
    run_plan_on_host(p, src, dst)
    assert np.array_equal(dst[0::2], src) and not dst[1::2].any()


def test_planner_r2r_rejects():
    x = np.zeros(8)
    y = np.zeros(8)
    with pytest.raises(ValueError):
        fa.plan_r2r_1d(1, x, y, fa.REDFT00)              # logical size 0 (reference.texi:806)
    with pytest.raises(ValueError):
        fa.plan_r2r_1d(8, x, y, 11)                      # not an fftw_r2r_kind
    with pytest.raises(ValueError):
        fa.plan_r2r_1d(0, x, y, fa.R2HC)
    with pytest.raises(ValueError):
        fa.plan_many_r2r(1, [8], -1, x, None, 1, 8, y, None, 1, 8, [fa.R2HC])


def test_planner_r2r_reference_verifier():
    """the reference's r2r self-test on the planner's step lists"""
    for kinds, shape in [([fa.REDFT10], (12,)), ([fa.RODFT11, fa.REDFT00], (5, 6)), ([fa.HC2R], (9,))]:
        n = int(np.prod(shape))

        def apply(x, shape=shape, kinds=kinds, n=n):
            v = x.shape[0]
            xin = np.ascontiguousarray(x.reshape(-1))
            y = np.zeros(v * n)
            p = fa.plan_many_r2r(len(shape), list(shape), v, xin, None, 1, n, y, None, 1, n, kinds)
            run_plan_on_host(p, xin, y)
            return y.reshape((v,) + tuple(shape))
        V.verify_r2r(apply, shape, kinds, vecn=2, rounds=2)


def test_r2r_print_plan():
    x = np.zeros(64)
    p = fa.plan_r2r_1d(64, x, x, fa.REDFT10)
    s = p.sprint()
    assert "rdft-r2r" in s and "r2r-pre-e10" in s and "+r2r-post-e10" in s


def _random_r2r_cases(seed, count, budget_total):
    """the reference's random sweep (fftw/tests/check.pl:186-251) restricted to r2r problems:
    random rank <= 3, dims from factors <= 13 (plus the odd prime), a random kind per dim,
    vector none / contiguous / interleaved, in or out of place"""
    rng = np.random.default_rng(seed)
    primes = [2, 3, 5, 7, 11, 13]
    for case in range(count):
        rank = int(rng.integers(1, 4))
        shape, kinds = [], []
        budget = budget_total ** (1.0 / rank)
        for _ in range(rank):
            n = 1
            while True:
                f = primes[int(rng.integers(0, len(primes)))]
                if n * f > max(2.0, budget):
                    break
                n *= f
                if rng.random() < 0.25:
                    break
            if rng.random() < 0.15:
                n = [17, 31, 37, 97, 101][int(rng.integers(0, 5))]     # generic / Rader / Bluestein inside r2r
            k = int(rng.integers(0, 11))
            if k == fa.REDFT00 and n < 2:
                n = 2
            shape.append(n)
            kinds.append(k)
        vtype = int(rng.integers(0, 3))
        v = 1 if vtype == 0 else int(rng.integers(2, 5))
        inplace = bool(rng.random() < 0.5)
        nn = int(np.prod(shape))
        if vtype == 2:
            x = rrand(rng, *(tuple(shape) + (v,)))
            stride, dist = v, 1
        else:
            x = rrand(rng, *((v,) + tuple(shape)))
            stride, dist = 1, nn
        yield case, shape, kinds, v, stride, dist, inplace, x


def test_planner_r2r_random_sweep():
    for case, shape, kinds, v, stride, dist, inplace, x in _random_r2r_cases(4321, 60, 20000):
        xin = x.reshape(-1).copy()
        y = xin if inplace else np.zeros_like(xin)
        p = fa.plan_many_r2r(len(shape), shape, v, xin, None, stride, dist, y, None, stride, dist, kinds)
        run_plan_on_host(p, xin, y)
        want = np.zeros(x.size)
        oracle_r2r(x.reshape(-1), shape, kinds, howmany=v, out=want, istride=stride, idist=dist,
                   ostride=stride, odist=dist)
        e = aerror(y, want)
        assert e <= TOL, (case, shape, [NAMES[k] for k in kinds], v, stride, inplace, e, p.sprint())


# ------------------------------------------------------------------ GPU tier

def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS, ids=NAMES)
def test_gpu_r2r_1d_vs_oracle(kind):
    import torch
    rng = np.random.default_rng(200 + kind)
    for n in sizes_for(kind, [1, 2, 3, 5, 8, 15, 16, 17, 64, 100, 127, 512, 1000, 1024, 4096, 4099, 30030]):
        hm = 3
        x = rrand(rng, hm * n)
        dx = _dev(x)
        dy = torch.zeros(hm * n, dtype=torch.float64, device="cuda")
        p = fa.plan_many_r2r(1, [n], hm, dx, None, 1, n, dy, None, 1, n, [kind])
        p.execute()
        p.sync()
        assert aerror(dy.cpu().numpy(), oracle_r2r(x, [n], [kind], howmany=hm)) <= TOL, (NAMES[kind], n)
        assert np.array_equal(dx.cpu().numpy(), x)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,kinds,howmany", [
    ([8, 12], [fa.REDFT10, fa.REDFT00], 3),
    ([7, 9], [fa.REDFT11, fa.RODFT11], 2),
    ([4, 5, 6], [fa.R2HC, fa.DHT, fa.RODFT01], 2),
    ([3, 1, 5], [fa.REDFT01, fa.RODFT10, fa.HC2R], 4),
    ([33, 20], [fa.RODFT00, fa.REDFT00], 2),
    ([256, 256], [fa.REDFT10, fa.REDFT10], 2),
    ([128, 96], [fa.RODFT01, fa.REDFT11], 1),
])
def test_gpu_r2r_multidim_vs_oracle(shape, kinds, howmany):
    import torch
    rng = np.random.default_rng(17)
    n = int(np.prod(shape))
    x = rrand(rng, howmany * n)
    dx = _dev(x)
    dy = torch.zeros(howmany * n, dtype=torch.float64, device="cuda")
    p = fa.plan_many_r2r(len(shape), shape, howmany, dx, None, 1, n, dy, None, 1, n, kinds)
    p.execute()
    p.sync()
    assert aerror(dy.cpu().numpy(), oracle_r2r(x, shape, kinds, howmany=howmany)) <= TOL


@pytest.mark.gpu
def test_gpu_r2r_strided_inplace_newarray():
    import torch
    rng = np.random.default_rng(18)
    shape, kinds, hm = [6, 10], [fa.REDFT10, fa.RODFT01], 3
    ie, oe, istr, ostr = [7, 12], [6, 11], 2, 3
    idist = ie[0] * ie[1] * istr + 5
    odist = oe[0] * oe[1] * ostr + 1
    x = rrand(rng, hm * idist)
    dx = _dev(x)
    dy = torch.full((hm * odist,), 7.0, dtype=torch.float64, device="cuda")
    p = fa.plan_many_r2r(2, shape, hm, dx, ie, istr, idist, dy, oe, ostr, odist, kinds)
    p.execute()
    p.sync()
    want = np.full(hm * odist, 7.0)
    oracle_r2r(x, shape, kinds, howmany=hm, out=want, inembed=ie, istride=istr, idist=idist,
               onembed=oe, ostride=ostr, odist=odist)
    got = dy.cpu().numpy()
    assert aerror(got, want) <= TOL
    assert np.array_equal(got == 7.0, want == 7.0)
    # new-array execution, in place, every kind
    for kind in KINDS:
        n, hm = 96, 50
        a = rrand(rng, hm * n)
        b = rrand(rng, hm * n)
        da, db = _dev(a), _dev(b)
        p = fa.plan_many_r2r(1, [n], hm, da, None, 1, n, da, None, 1, n, [kind])
        p.execute()
        p.execute_r2r(db, db)
        p.sync()
        assert aerror(da.cpu().numpy(), oracle_r2r(a, [n], [kind], howmany=hm)) <= TOL
        assert aerror(db.cpu().numpy(), oracle_r2r(b, [n], [kind], howmany=hm)) <= TOL


@pytest.mark.gpu
def test_gpu_r2r_random_sweep():
    import torch
    for case, shape, kinds, v, stride, dist, inplace, x in _random_r2r_cases(987, 80, 200000):
        dx = _dev(x.reshape(-1))
        dy = dx if inplace else torch.zeros_like(dx)
        p = fa.plan_many_r2r(len(shape), shape, v, dx, None, stride, dist, dy, None, stride, dist, kinds)
        p.execute()
        p.sync()
        want = np.zeros(x.size)
        oracle_r2r(x.reshape(-1), shape, kinds, howmany=v, out=want, istride=stride, idist=dist,
                   ostride=stride, odist=dist)
        e = aerror(dy.cpu().numpy(), want)
        assert e <= TOL, (case, shape, [NAMES[k] for k in kinds], v, stride, inplace, e, p.sprint())


@pytest.mark.gpu
def test_gpu_r2r_is_capturable_in_a_hip_graph():
    import torch
    rng = np.random.default_rng(21)
    n, b = 4096, 64
    x0, x1 = rrand(rng, b * n), rrand(rng, b * n)
    xd = _dev(x0)
    yd = torch.zeros_like(xd)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        p = fa.plan_many_r2r(1, [n], b, xd, None, 1, n, yd, None, 1, n, [fa.REDFT10])
        p.execute()                       # warm-up: tables and scratch exist before capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            p.execute()
    xd.copy_(torch.from_numpy(x1))
    g.replay()
    torch.cuda.synchronize()
    assert aerror(yd.cpu().numpy(), oracle_r2r(x1, [n], [fa.REDFT10], howmany=b)) <= TOL


@pytest.mark.gpu
def test_gpu_r2r_host_pointers():
    rng = np.random.default_rng(19)
    n = 200
    x = rrand(rng, n)
    y = np.zeros(n)
    p = fa.plan_r2r_1d(n, x, y, fa.REDFT10)
    p.execute()
    p.sync()
    assert aerror(y, oracle_r2r(x, [n], [fa.REDFT10])) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS, ids=NAMES)
def test_gpu_r2r_inverse_pairs_large(kind):
    """kind^-1(kind(x)) = N x with N the logical size (reference.texi:2135-2150),
    at sizes beyond the oracle; also chunked (the batch does not fit one chunk)."""
    import torch
    if kind == fa.HC2R:
        pytest.skip("HC2R ignores the imaginary slots of DC / Nyquist; covered through R2HC -> HC2R")
    n, hm = 1 << 18, 24
    g = torch.Generator(device="cuda").manual_seed(kind)
    x = torch.rand(hm * n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    y = torch.zeros_like(x)
    z = torch.zeros_like(x)
    fa.set_chunk_bytes(64 << 20)
    try:
        p = fa.plan_many_r2r(1, [n], hm, x, None, 1, n, y, None, 1, n, [kind])
        q = fa.plan_many_r2r(1, [n], hm, y, None, 1, n, z, None, 1, n, [INVERSE[kind]])
    finally:
        fa.set_chunk_bytes(0)
    p.execute()
    q.execute()
    q.sync()
    N = logical_n(kind, n)
    err = (z / N - x).abs().max().item() / x.abs().max().item()
    assert err <= TOL, (NAMES[kind], err)


@pytest.mark.gpu
def test_gpu_r2r_r2hc_equals_r2c_large():
    """R2HC is the r2c half spectrum in halfcomplex order"""
    import torch
    n, hm = 1 << 20, 8
    x = torch.rand(hm * n, dtype=torch.float64, device="cuda") - 0.5
    hc = torch.zeros_like(x)
    spec = torch.zeros(hm * (n // 2 + 1), dtype=torch.complex128, device="cuda")
    p = fa.plan_many_r2r(1, [n], hm, x, None, 1, n, hc, None, 1, n, [fa.R2HC])
    q = fa.plan_many_dft_r2c(1, [n], hm, x, None, 1, n, spec, None, 1, n // 2 + 1)
    p.execute()
    q.execute()
    q.sync()
    p.sync()
    hc = hc.view(hm, n)
    spec = spec.view(hm, n // 2 + 1)
    # the two plans end in different kernels (untangle with the R2HC epilogue / streaming untangle): the same
    # arithmetic, but not necessarily the same FMA contractions -- equal to rounding, not bit for bit
    scale = spec.abs().max().item()
    assert (hc[:, :n // 2 + 1] - spec.real).abs().max().item() <= 8e-16 * scale
    assert (hc[:, n // 2 + 1:] - torch.flip(spec.imag[:, 1:n // 2], dims=[1])).abs().max().item() <= 8e-16 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("n", [243, 1 << 16])
def test_gpu_r2r_r2hc_equals_r2c_bitwise_when_the_kernels_are_the_same(monkeypatch, n):
    """Where R2HC and r2c run the SAME kernels -- the unfused plans (FFTW_AMD_R2R_UNFUSED=1: r2c steps, then a
    separate r2r step that only reorders), and odd lengths, which are never fused -- the halfcomplex output
    is the r2c half spectrum bit for bit.  (The fused large-n plans end in different kernels -- the radix-4
    untangle with the R2HC store hook against the streaming untangle -- whose FMA contractions differ: they
    are compared to rounding in test_gpu_r2r_r2hc_equals_r2c_large.)"""
    import torch
    monkeypatch.setenv("FFTW_AMD_R2R_UNFUSED", "1")
    hm = 5
    x = torch.rand(hm * n, dtype=torch.float64, device="cuda") - 0.5
    hc = torch.zeros_like(x)
    spec = torch.zeros(hm * (n // 2 + 1), dtype=torch.complex128, device="cuda")
    p = fa.plan_many_r2r(1, [n], hm, x, None, 1, n, hc, None, 1, n, [fa.R2HC])
    q = fa.plan_many_dft_r2c(1, [n], hm, x, None, 1, n, spec, None, 1, n // 2 + 1)
    p.execute()
    q.execute()
    q.sync()
    p.sync()
    hc = hc.view(hm, n)
    spec = spec.view(hm, n // 2 + 1)
    assert torch.equal(hc[:, :n // 2 + 1], spec.real.contiguous())
    assert torch.equal(hc[:, n // 2 + 1:], torch.flip(spec.imag[:, 1:(n + 1) // 2], dims=[1]).contiguous())


@pytest.mark.gpu
def test_gpu_r2r_reference_verifier():
    import torch
    for kinds, shape in [([fa.REDFT10], (1000,)), ([fa.RODFT11, fa.REDFT00], (50, 60)),
                         ([fa.HC2R], (243,)), ([fa.DHT, fa.R2HC], (32, 31)), ([fa.RODFT00], (4095,)),
                         ([fa.REDFT01, fa.RODFT10, fa.REDFT11], (8, 9, 10))]:
        n = int(np.prod(shape))

        def apply(x, shape=shape, kinds=kinds, n=n):
            v = x.shape[0]
            dx = _dev(x.reshape(-1))
            dy = torch.zeros(v * n, dtype=torch.float64, device="cuda")
            p = fa.plan_many_r2r(len(shape), list(shape), v, dx, None, 1, n, dy, None, 1, n, kinds)
            p.execute()
            p.sync()
            return dy.cpu().numpy().reshape((v,) + tuple(shape))
        V.verify_r2r(apply, shape, kinds, vecn=2, rounds=3)


@pytest.mark.gpu
def test_gpu_r2r_measure_mode_and_wisdom():
    """FFTW_MEASURE on an r2r problem times the candidate configurations, records wisdom keyed
    by the kinds, and the measured plan is still correct"""
    import torch
    rng = np.random.default_rng(33)
    n, hm = 1 << 14, 64
    x = rrand(rng, hm * n)
    fa.forget_wisdom()
    dx = _dev(x)
    dy = torch.zeros_like(dx)
    p = fa.plan_many_r2r(1, [n], hm, dx, None, 1, n, dy, None, 1, n, [fa.RODFT10], fa.MEASURE)
    w = fa.export_wisdom_to_string()
    assert "k9." in w, w                      # the wisdom key carries the r2r kind
    dx.copy_(torch.from_numpy(x))             # MEASURE overwrote the arrays
    p.execute()
    p.sync()
    assert aerror(dy.cpu().numpy(), oracle_r2r(x, [n], [fa.RODFT10], howmany=hm)) <= TOL
    q = fa.plan_many_r2r(1, [n], hm, dx, None, 1, n, dy, None, 1, n, [fa.RODFT10], fa.WISDOM_ONLY)
    q.execute()
    q.sync()
    assert aerror(dy.cpu().numpy(), oracle_r2r(x, [n], [fa.RODFT10], howmany=hm)) <= TOL
    fa.forget_wisdom()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [(fa.R2HC, 1024), (fa.DHT, 512), (fa.REDFT10, 2048), (fa.RODFT10, 256),
                                    (fa.REDFT00, 513), (fa.RODFT00, 511), (fa.REDFT10, 128)])
def test_gpu_r2r_fused_rows_epilogues(kind, n):
    """short contiguous r2r rows: the inner real transform and the r2r post-processing run in the
    fused rows kernel (one trip, or shuffle + one trip); ragged batch, also as the last axis of 2-D"""
    import torch
    rng = np.random.default_rng(n + kind)
    hm = 37
    x = rrand(rng, hm * n)
    dx = _dev(x)
    dy = torch.zeros_like(dx)
    p = fa.plan_many_r2r(1, [n], hm, dx, None, 1, n, dy, None, 1, n, [kind])
    assert "r2c-rows+r2r-post" in p.sprint(), p.sprint()
    p.execute()
    p.sync()
    assert aerror(dy.cpu().numpy(), oracle_r2r(x, [n], [kind], howmany=hm)) <= TOL
    shape, kinds = [24, n], [fa.REDFT01, kind]
    x = rrand(rng, 2 * 24 * n)
    dx = _dev(x)
    dy = torch.zeros_like(dx)
    p = fa.plan_many_r2r(2, shape, 2, dx, None, 1, 24 * n, dy, None, 1, 24 * n, kinds)
    assert "r2c-rows+r2r-post" in p.sprint(), p.sprint()
    p.execute()
    p.sync()
    assert aerror(dy.cpu().numpy(), oracle_r2r(x, shape, kinds, howmany=2)) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [(fa.HC2R, 1024), (fa.REDFT01, 2048), (fa.RODFT01, 256), (fa.HC2R, 128)])
def test_gpu_r2r_fused_rows_prologues(kind, n):
    """the transposed kinds: the r2r pre-processing runs inside the fused c2r rows kernel"""
    import torch
    rng = np.random.default_rng(n + kind)
    hm = 37
    x = rrand(rng, hm * n)
    dx = _dev(x)
    dy = torch.zeros_like(dx)
    p = fa.plan_many_r2r(1, [n], hm, dx, None, 1, n, dy, None, 1, n, [kind])
    assert "c2r-rows+r2r-pre" in p.sprint(), p.sprint()
    p.execute()
    p.sync()
    assert aerror(dy.cpu().numpy(), oracle_r2r(x, [n], [kind], howmany=hm)) <= TOL
    assert np.array_equal(dx.cpu().numpy(), x)
    shape, kinds = [20, n], [fa.RODFT10, kind]
    x = rrand(rng, 2 * 20 * n)
    dx = _dev(x)
    p = fa.plan_many_r2r(2, shape, 2, dx, None, 1, 20 * n, dx, None, 1, 20 * n, kinds)     # in place
    assert "c2r-rows+r2r-pre" in p.sprint(), p.sprint()
    p.execute()
    p.sync()
    assert aerror(dx.cpu().numpy(), oracle_r2r(x, shape, kinds, howmany=2)) <= TOL


@pytest.mark.parametrize("kind,n", [(fa.REDFT01, 1024), (fa.RODFT01, 512), (fa.REDFT01, 128)])
def test_planner_r2r_rows_store_shuffle(kind, n):
    """DCT-III / DST-III short rows: one fused launch (prologue, backward pass, output shuffle in the
    store); checked through the numpy step interpreter, also with a strided destination"""
    from step_interp import run_plan_on_host
    rng = np.random.default_rng(n + kind)
    hm = 5
    x = rrand(rng, hm * n)
    y = np.zeros(hm * n)
    p = fa.plan_many_r2r(1, [n], hm, x, None, 1, n, y, None, 1, n, [kind])
    assert "c2r-rows+r2r-pre+post" in p.sprint() and len(p.steps()) == 1, p.sprint()
    run_plan_on_host(p, x, y)
    assert aerror(y, oracle_r2r(x, [n], [kind], howmany=hm)) <= TOL
    y = np.zeros(3 * hm * n)
    p = fa.plan_many_r2r(1, [n], hm, x, None, 1, n, y, None, 3, 3 * n, [kind])
    assert "c2r-rows+r2r-pre+post" in p.sprint(), p.sprint()
    run_plan_on_host(p, x, y)
    assert aerror(y[::3], oracle_r2r(x, [n], [kind], howmany=hm)) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [(fa.REDFT01, 1024), (fa.RODFT01, 2048), (fa.REDFT01, 128), (fa.RODFT01, 256)])
def test_gpu_r2r_rows_store_shuffle(kind, n):
    """the fused c2r rows kernel also does the DCT-III / DST-III output shuffle: ragged batch,
    in place, strided destination"""
    import torch
    rng = np.random.default_rng(n + kind)
    hm = 37
    x = rrand(rng, hm * n)
    dx = _dev(x)
    dy = torch.zeros_like(dx)
    p = fa.plan_many_r2r(1, [n], hm, dx, None, 1, n, dy, None, 1, n, [kind])
    assert "c2r-rows+r2r-pre+post" in p.sprint(), p.sprint()
    p.execute()
    p.sync()
    want = oracle_r2r(x, [n], [kind], howmany=hm)
    assert aerror(dy.cpu().numpy(), want) <= TOL
    assert np.array_equal(dx.cpu().numpy(), x)
    dz = torch.zeros(3 * hm * n, dtype=torch.float64, device="cuda")
    p = fa.plan_many_r2r(1, [n], hm, dx, None, 1, n, dz, None, 3, 3 * n, [kind])
    p.execute()
    p.sync()
    assert aerror(dz.cpu().numpy()[::3], want) <= TOL
    p = fa.plan_many_r2r(1, [n], hm, dx, None, 1, n, dx, None, 1, n, [kind])      # in place
    p.execute()
    p.sync()
    assert aerror(dx.cpu().numpy(), want) <= TOL
