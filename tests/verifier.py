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
