"""Restatement of the reference's self-checking FFT verifier.

Reference: fftw/libbench2/verify-lib.c (after Ergun 1995) -- impulse and
constant input (verify-lib.c:284-325), linearity (:327-356), time-shift and
frequency-shift along every dimension (:360-414), relative L-infinity error
(:36-58), 10 rounds at tolerance 1e-10 for double (bench-main.c:62,70).

`apply(x)` must compute the transform under test on a complex128 array of
shape (vecn,) + shape and return an array of the same shape.  The checks are
size-independent properties, so they pin a transform at sizes where no stored
answer exists.  These are the known-answer tests the reference's own test
program holds for the codelet path; they pin both the CPU oracle
(tests/test_oracle_pin.py) and the GPU path (tests/test_gpu_parity.py).
"""
import numpy as np

from util import aerror, crand

ROUNDS = 10
TOL = 1e-10


def _chk(got, want, what, tol):
    e = aerror(got, want)
    assert e <= tol, "%s: relative error %.3e > %.1e" % (what, e, tol)
    return e


def impulse(apply, shape, vecn, rng, rounds=ROUNDS, tol=TOL):
    """delta -> constant and constant -> delta, directly and through random splits A = B + C"""
    n = int(np.prod(shape))
    worst = 0.0
    for mode in ("impulse", "constant"):
        inA = np.zeros((vecn, n), dtype=np.complex128)
        outA = np.zeros((vecn, n), dtype=np.complex128)
        for i in range(vecn):
            if mode == "impulse":
                x = np.sqrt(n) * (i + 1) / (vecn + 1.0)
                inA[i, 0] = x
                outA[i, :] = x
            else:
                x = (i + 1) / ((vecn + 1.0) * np.sqrt(n))
                inA[i, :] = x
                outA[i, 0] = n * x
        worst = max(worst, _chk(apply(inA.reshape((vecn,) + shape)).reshape(vecn, n), outA,
                                mode + " 1", tol))
        for _ in range(rounds):
            inB = crand(rng, vecn, n)
            inC = inA - inB
            s = apply(inB.reshape((vecn,) + shape)) + apply(inC.reshape((vecn,) + shape))
            worst = max(worst, _chk(s.reshape(vecn, n), outA, mode, tol))
    return worst


def linear(apply, shape, vecn, rng, rounds=ROUNDS, tol=TOL, realp=False):
    worst = 0.0
    full = (vecn,) + shape
    for _ in range(rounds):
        a = crand(rng, 1)[0]
        b = crand(rng, 1)[0]
        if realp:
            a, b = a.real, b.real
        inA, inB = crand(rng, *full), crand(rng, *full)
        if realp:
            inA, inB = inA.real + 0j, inB.real + 0j
        want = a * apply(inA) + b * apply(inB)
        got = apply(a * inA + b * inB)
        worst = max(worst, _chk(got, want, "linear", tol))
    return worst


def _phase(shape, dim, sign):
    """exp(sign * 2 pi i k / n_dim) along `dim`, broadcastable over (vecn,)+shape"""
    n = shape[dim]
    ph = np.exp(sign * 2j * np.pi * np.arange(n) / n)
    sh = [1] * (len(shape) + 1)
    sh[dim + 1] = n
    return ph.reshape(sh)


def time_shift(apply, shape, vecn, rng, sign=-1, rounds=ROUNDS, tol=TOL):
    """x rotated by one sample along a dimension <-> output times a phase ramp"""
    worst = 0.0
    full = (vecn,) + shape
    for dim in range(len(shape)):
        for _ in range(rounds):
            inA = crand(rng, *full)
            inB = np.roll(inA, 1, axis=dim + 1)       # B[j] = A[j-1]
            outA, outB = apply(inA), apply(inB)
            # forward (sign -1): F(B)[k] = F(A)[k] e^{-2 pi i k / n}
            want = outB * _phase(shape, dim, -sign)
            worst = max(worst, _chk(want, outA, "time shift", tol))
    return worst


def freq_shift(apply, shape, vecn, rng, sign=-1, rounds=ROUNDS, tol=TOL):
    """input times a phase ramp <-> output rotated by one bin"""
    worst = 0.0
    full = (vecn,) + shape
    for dim in range(len(shape)):
        for _ in range(rounds):
            inA = crand(rng, *full)
            inB = inA * _phase(shape, dim, -sign)      # forward: times e^{+2 pi i j / n}
            outA, outB = apply(inA), apply(inB)
            want = np.roll(outB, -1, axis=dim + 1)     # F(B)[k] = F(A)[k-1]
            worst = max(worst, _chk(want, outA, "freq shift", tol))
    return worst


def verify_c2c(apply, shape, vecn=1, sign=-1, seed=1, rounds=ROUNDS, tol=TOL):
    """the battery the reference runs for a complex DFT (verify-dft.c:103-153)"""
    rng = np.random.default_rng(seed)
    shape = tuple(shape)
    e = [impulse(apply, shape, vecn, rng, rounds, tol),
         linear(apply, shape, vecn, rng, rounds, tol),
         time_shift(apply, shape, vecn, rng, sign, rounds, tol),
         freq_shift(apply, shape, vecn, rng, sign, rounds, tol)]
    return max(e)


# ---------------------------------------------------------------------------
# r2r: restatement of fftw/libbench2/verify-r2r.c (impulse response :404-418 and
# :455-496, linearity :431-453, time shift with the per-kind boundary rules of
# rarolr :201-312 and phase factors :314-328, per-kind parameters :565-632).

(R2HC, HC2R, DHT, REDFT00, REDFT01, REDFT10, REDFT11,
 RODFT00, RODFT01, RODFT10, RODFT11) = range(11)


def _cos00(i, j, n):
    return np.cos(2.0 * np.pi * ((i * j) % n) / n)


def _sin00(i, j, n):
    return np.sin(2.0 * np.pi * ((i * j) % n) / n)


_TRIG = {
    "cos00": _cos00,
    "cos01": lambda i, j, n: _cos00(i, 2 * j + 1, 2 * n),
    "cos10": lambda i, j, n: _cos00(2 * i + 1, j, 2 * n),
    "cos11": lambda i, j, n: _cos00(2 * i + 1, 2 * j + 1, 4 * n),
    "sin00": _sin00,
    "sin01": lambda i, j, n: _sin00(i, 2 * j + 1, 2 * n),
    "sin10": lambda i, j, n: _sin00(2 * i + 1, j, 2 * n),
    "sin11": lambda i, j, n: _sin00(2 * i + 1, 2 * j + 1, 4 * n),
    "realhalf": lambda i, j, n: np.where(j <= n - j, 1.0, 0.0),
    "coshalf": lambda i, j, n: np.where(j <= n - j, _cos00(i, j, n), _cos00(i, n - j, n)),
    "unity": lambda i, j, n: np.ones_like(j, dtype=np.float64),
}

# kind -> (i0, k0, impulse trig, amplitude factor, shift trig)
_R2R_DIM = {
    R2HC: (0, 0, "realhalf", 1.0, "coshalf"),
    DHT: (0, 0, "unity", 1.0, "cos00"),
    HC2R: (0, 0, "unity", 1.0, "cos00"),
    REDFT00: (0, 0, "cos00", 1.0, "cos00"),
    REDFT01: (0, 0, "cos01", 1.0, "cos01"),
    REDFT10: (0, 0, "cos10", 2.0, "cos00"),
    REDFT11: (0, 0, "cos11", 2.0, "cos01"),
    RODFT00: (1, 1, "sin00", 2.0, "cos00"),
    RODFT01: (1, 0, "sin01", 2.0, "cos01"),
    RODFT10: (0, 1, "sin10", 2.0, "cos00"),
    RODFT11: (0, 0, "sin11", 2.0, "cos01"),
}


def _logical_n(kind, n):
    if kind <= DHT:
        return n
    return 2 * (n + (-1 if kind == REDFT00 else (1 if kind == RODFT00 else 0)))


def _rarolr(a, axis, kind):
    """B = rotate-left A + rotate-right A along `axis` with the boundary rules of the kind"""
    a = np.moveaxis(a, axis, -1)
    n = a.shape[-1]
    b = np.zeros_like(a)
    b[..., :n - 1] = a[..., 1:]
    if kind in (DHT, R2HC):
        b[..., n - 1] = a[..., 0]
        b[..., 0] += a[..., n - 1]
    elif kind == HC2R:
        if n > 2:
            h = n // 2
            b[..., n - 1] = 0.0
            b[..., 0] += a[..., 1]
            if n % 2 == 0:
                b[..., h] += a[..., h - 1] - a[..., h + 1]
                b[..., h + 1] += -a[..., h]
            else:
                b[..., h] += a[..., h] - a[..., h + 1]
                b[..., h + 1] += -a[..., h + 1] - a[..., h]
        else:
            b[..., n - 1] = a[..., 0]
            b[..., 0] += a[..., n - 1]
    else:
        L0, L1, R0, R1 = {
            REDFT00: (0, 1, 0, 1), REDFT01: (0, 1, 0, 0), REDFT10: (1, 0, 1, 0),
            REDFT11: (1, 0, -1, 0), RODFT00: (0, 0, 0, 0), RODFT01: (0, 0, 0, 1),
            RODFT10: (-1, 0, -1, 0), RODFT11: (-1, 0, 1, 0),
        }[kind]
        b[..., n - 1] = R0 * a[..., n - 1] + (R1 * a[..., n - 2] if n > 1 else 0.0)
        b[..., 0] += L0 * a[..., 0] + (L1 * a[..., 1] if n > 1 else 0.0)
    b[..., 1:] += a[..., :n - 1]
    return np.moveaxis(b, -1, axis)


def verify_r2r(apply, shape, kinds, vecn=1, seed=1, rounds=ROUNDS, tol=TOL):
    """apply(x): r2r of kinds[d] along dim d of a float64 array of shape (vecn,) + shape"""
    from util import rrand
    rng = np.random.default_rng(seed)
    shape = tuple(shape)
    full = (vecn,) + shape
    worst = 0.0
    # impulse at the origin of every vector
    resp = np.ones(shape, dtype=np.float64)
    for d, (n, k) in enumerate(zip(shape, kinds)):
        i0, k0, ti, amp, _ = _R2R_DIM[k]
        if k == RODFT01 and n == 1:
            amp = 1.0
        j = np.arange(n, dtype=np.int64)
        t = amp * _TRIG[ti](i0, k0 + j, _logical_n(k, n))
        resp = resp * t.reshape([n if e == d else 1 for e in range(len(shape))])
    inA = np.zeros(full)
    outA = np.zeros(full)
    for i in range(vecn):
        x = (i + 1) / (vecn + 1.0)
        inA[(i,) + (0,) * len(shape)] = x
        outA[i] = x * resp
    worst = max(worst, _chk(apply(inA), outA, "impulse 1", tol))
    for _ in range(rounds):
        inB = rrand(rng, *full)
        worst = max(worst, _chk(apply(inB) + apply(inA - inB), outA, "impulse", tol))
    # linearity
    for _ in range(rounds):
        al, be = rrand(rng, 2)
        a, b = rrand(rng, *full), rrand(rng, *full)
        worst = max(worst, _chk(apply(al * a + be * b), al * apply(a) + be * apply(b), "linear", tol))
    # time shift along every dim
    for d, (n, k) in enumerate(zip(shape, kinds)):
        _, k0, _, _, ts = _R2R_DIM[k]
        if n == 1:
            continue          # nothing to shift; both sides are 0 up to rounding
        j = np.arange(n, dtype=np.int64)
        ph = 2.0 * _TRIG[ts](1, j + k0, _logical_n(k, n))
        ph = ph.reshape([1] + [n if e == d else 1 for e in range(len(shape))])
        for _ in range(rounds):
            a = rrand(rng, *full)
            b = _rarolr(a, d + 1, k)
            worst = max(worst, _chk(apply(a) * ph, apply(b), "time shift", tol))
    return worst
