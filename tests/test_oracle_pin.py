"""Pins the CPU oracle (oracle/fftw_oracle.c) before anything is compared to it.

The reference stores no golden vectors; what its own test program holds for
the codelet path is the self-checking verifier (fftw/libbench2/verify-lib.c)
run over the sweeps of fftw/tests/check.pl:126-184 at tolerance 1e-10.  The
oracle must pass that battery, must match the committed long-double fixtures
(tests/golden/, inputs = the verifier's own drand48 sequence), and must agree
with closed-form answers.  numpy's pocketfft is used as an independent
cross-check only.
"""
import os

import numpy as np
import pytest

import verifier
from util import TOL, aerror, crand, oracle, oracle_c2r, oracle_dft, oracle_r2c, rrand

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _apply_oracle(shape, sign=-1):
    def apply(x):
        vecn = x.shape[0]
        return oracle_dft(x, shape, vecn, sign).reshape(x.shape)
    return apply


# sizes 1..100 and 2^7..2^12 as in the reference sweep (check.pl:126-173)
SWEEP = list(range(1, 101)) + [128, 256, 512, 1024, 2048, 4096]


@pytest.mark.parametrize("n", SWEEP)
def test_oracle_passes_reference_verifier_1d(n):
    for sign in (-1, +1):
        e = verifier.verify_c2c(_apply_oracle((n,), sign), (n,), vecn=1, sign=sign, rounds=3)
        assert e <= TOL


@pytest.mark.parametrize("shape", [(4, 4), (8, 16), (13, 11), (30, 30), (5, 6, 7), (2, 3, 4, 5)])
def test_oracle_passes_reference_verifier_nd(shape):
    e = verifier.verify_c2c(_apply_oracle(shape), shape, vecn=2, rounds=2)
    assert e <= TOL


@pytest.mark.parametrize("n", [173, 257, 1009, 1031, 17 * 19, 65537, 15015, 17408, 1 << 16])
def test_oracle_prime_paths_against_properties(n):
    """generic / Rader / Bluestein sizes"""
    e = verifier.verify_c2c(_apply_oracle((n,)), (n,), vecn=1, rounds=1)
    assert e <= TOL


def test_oracle_matches_golden_c2c():
    z = np.load(os.path.join(GOLD, "c2c_1d.npz"))
    sizes = sorted(int(k[1:-3]) for k in z.files if k.startswith("n") and k.endswith("_in"))
    assert 15015 in sizes and 1031 in sizes
    for n in sizes:
        x = z["n%d_in" % n]
        assert aerror(oracle_dft(x, (n,), 1, -1), z["n%d_fwd" % n]) < 1e-13, n
        assert aerror(oracle_dft(x, (n,), 1, +1), z["n%d_bwd" % n]) < 1e-13, n
    for n in (4, 13, 64, 1024):
        x = z["v3n%d_in" % n]
        # N*V: contiguous vectors
        assert aerror(oracle_dft(x, (n,), 3).reshape(3, n), z["v3n%d_fwd" % n]) < 1e-13
        # NvV: interleaved vectors (stride 3, dist 1)
        xi = np.ascontiguousarray(x.T)
        out = oracle_dft(xi, (n,), 3, istride=3, idist=1, ostride=3, odist=1).reshape(n, 3)
        assert aerror(out.T, z["v3n%d_fwd" % n]) < 1e-13


def test_oracle_matches_golden_real_and_nd():
    z = np.load(os.path.join(GOLD, "r2c_1d.npz"))
    for k in [f for f in z.files if f.endswith("_in")]:
        n = int(k[1:-3])
        x, y = z[k], z["n%d_out" % n]
        assert aerror(oracle_r2c(x, (n,)), y) < 1e-13, n
        assert aerror(oracle_c2r(y, (n,)), x * n) < 1e-13, n
    z = np.load(os.path.join(GOLD, "nd.npz"))
    for k in [f for f in z.files if f.startswith("c") and f.endswith("_in")]:
        shape = tuple(int(s) for s in k[1:-3].split("x"))
        assert aerror(oracle_dft(z[k], shape).reshape(shape), z[k[:-3] + "_fwd"]) < 1e-13
    for k in [f for f in z.files if f.startswith("r") and f.endswith("_in")]:
        shape = tuple(int(s) for s in k[1:-3].split("x"))
        hs = (shape[0], shape[1] // 2 + 1)
        y = z[k[:-3] + "_out"]
        assert aerror(oracle_r2c(z[k], shape).reshape(hs), y) < 1e-13
        assert aerror(oracle_c2r(y, shape).reshape(shape), z[k] * shape[0] * shape[1]) < 1e-13


@pytest.mark.parametrize("n", [8, 60, 97, 1024, 1031, 6000])
def test_oracle_single_tone_closed_form(n):
    """x[j] = exp(2 pi i f j / n)  ->  n at bin f, 0 elsewhere"""
    for f in (0, 1, n // 3, n - 1):
        x = np.exp(2j * np.pi * ((f * np.arange(n)) % n) / n)
        y = oracle_dft(x, (n,))
        want = np.zeros(n, dtype=complex)
        want[f] = n
        assert np.abs(y - want).max() < 1e-9 * n


def test_oracle_cexp_octant_symmetries():
    """real_cexp restatement: exact symmetries the octant reduction guarantees"""
    import ctypes as C
    out = (C.c_double * 2)()
    lib = oracle()
    n = 1 << 20
    for m in (0, 1, 12345, n // 8, n // 4, n // 2, 3 * n // 4, n - 1):
        lib.oracle_cexp(m, n, out)
        c, s = out[0], out[1]
        lib.oracle_cexp(n - m, n, out)
        assert out[0] == c and out[1] == -s           # conjugate symmetry, bit exact
        lib.oracle_cexp(m + n // 4, n, out)
        assert out[0] == -s and out[1] == c           # quarter-turn rotation, bit exact
        assert abs(c - np.cos(2 * np.pi * m / n)) < 2e-16 and abs(s - np.sin(2 * np.pi * m / n)) < 2e-16


def test_oracle_vs_numpy_crosscheck_and_real():
    rng = np.random.default_rng(5)
    for n in (6, 15, 64, 100, 1009, 4096, 10000):
        x = crand(rng, n)
        assert aerror(oracle_dft(x, (n,)), np.fft.fft(x)) < 1e-13
        xr = rrand(rng, n)
        y = oracle_r2c(xr, (n,))
        assert aerror(y, np.fft.rfft(xr)) < 1e-13
        assert y[0].imag == 0.0 and (n % 2 or y[n // 2].imag == 0.0)   # A.c:7155-7156
        assert aerror(oracle_c2r(y, (n,)), xr * n) < 1e-13


def _pins2():
    return np.load(os.path.join(GOLD, "pins2.npz"))


def test_oracle_matches_second_golden_set():
    """round 3 pins (tests/golden/make_golden2.py: defining sums in 80-bit long double, verifier inputs):
    Rader 65537 / 12289, Bluestein 8191, 13-smooth 60060 (sampled bins), odd-length r2c / c2r, 3-D c2c and
    r2c, and every r2r kind at an even and an odd length -- the sizes that until now were only checked
    oracle-vs-GPU"""
    from util import oracle_r2r
    z = _pins2()
    for n in (65537, 12289, 8191, 60060):
        x, bins = z["c%d_in" % n], z["c%d_bins" % n]
        scale = np.abs(z["c%d_fwd" % n]).max()
        assert np.abs(oracle_dft(x, (n,), 1, -1)[bins] - z["c%d_fwd" % n]).max() < 1e-13 * scale, n
        assert np.abs(oracle_dft(x, (n,), 1, +1)[bins] - z["c%d_bwd" % n]).max() < 1e-13 * scale, n
    for n in (77, 1001):
        x, y = z["r%d_in" % n], z["r%d_out" % n]
        assert aerror(oracle_r2c(x, (n,)), y) < 1e-13, n
        assert aerror(oracle_c2r(y, (n,)), x * n) < 1e-13, n
    for key in ("6x10x8", "5x6x7"):
        shape = tuple(int(v) for v in key.split("x"))
        assert aerror(oracle_dft(z["c3_%s_in" % key], shape).reshape(shape), z["c3_%s_fwd" % key]) < 1e-13
        hs = shape[:2] + (shape[2] // 2 + 1,)
        assert aerror(oracle_r2c(z["r3_%s_in" % key], shape).reshape(hs), z["r3_%s_out" % key]) < 1e-13
        assert aerror(oracle_c2r(z["r3_%s_out" % key], shape).reshape(shape), z["r3_%s_in" % key] * np.prod(shape)) < 1e-13
    for n in (16, 15, 1000, 243):
        for kind in range(11):
            got = oracle_r2r(z["k%d_n%d_in" % (kind, n)], (n,), [kind])
            assert aerror(got, z["k%d_n%d_out" % (kind, n)]) < 1e-13, (kind, n)


def test_oracle_generic_prime_path_against_the_direct_sum():
    """the oracle's O(n^2) prime solver (fftw_oracle.c dft_generic, restating A.c:3390-3448: folded
    x[j] +- x[n-j] Hartley-like sums) and the product's BflyOdd butterflies use the same symmetric
    formula; this check is structurally different -- the plain defining sum, every term computed on
    its own in long double with the angle reduced exactly -- for every odd prime the reference's generic
    solver serves (p < 173, A.h:1108-1114) and the next few beyond it"""
    from golden.make_golden import dft_ld
    rng2 = np.random.default_rng(173)
    primes = [p for p in range(3, 200) if all(p % q for q in range(2, int(p ** 0.5) + 1))]
    assert 167 in primes and 173 in primes
    for p in primes:
        x = (rng2.random(p) - 0.5) + 1j * (rng2.random(p) - 0.5)
        for sign in (-1, 1):
            want = dft_ld(x, sign).astype(np.complex128)
            assert aerror(oracle_dft(x, (p,), 1, sign), want) < 1e-14, p
