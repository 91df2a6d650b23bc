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
