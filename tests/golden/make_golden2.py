"""Second set of golden fixtures (round 3): one size per product code path that round 2 checked
only oracle-vs-GPU -- Rader (65537, 12289), Bluestein beyond one kernel (8191), mixed 13-smooth
(60060), odd-length r2c / c2r (77, 1001), 3-D shapes, and every r2r kind at one even and one odd
length.  Same recipe as make_golden.py: inputs = the reference verifier's drand48() - 0.5 sequence
after srand48(1) (fftw/libbench2/verify-lib.c:64-67), expected outputs = the DEFINING sums
evaluated directly in 80-bit long double with exactly reduced angles -- the DFT for the complex
and real transforms, and for the r2r kinds the formulas of the FFTW manual
(/root/reference/fftw/doc/reference.texi:2058-2230: R2HC / HC2R halfcomplex, DHT, REDFT00/01/10/11,
RODFT00/01/10/11, all unnormalised).  Nothing here comes from the oracle or from the product.

The long transforms store the whole input and a SAMPLE of output bins (`*_bins` + `*_vals`): a
direct O(n^2) evaluation of every bin of n = 65537 in long double takes minutes and adds nothing.

Run:  python tests/golden/make_golden2.py      (rewrites pins2.npz next to it)
"""
import os

import numpy as np

from make_golden import Drand48, dft_ld

HERE = os.path.dirname(os.path.abspath(__file__))
LD = np.longdouble
PI = LD(2) * np.arctan2(LD(1), LD(0))          # pi in long double


def dft_bins_ld(x, bins, sign=-1):
    """X[k] for k in bins, direct sum in long double, angles reduced exactly (j k mod n)"""
    x = np.asarray(x, dtype=np.clongdouble)
    n = x.shape[0]
    j = np.arange(n, dtype=np.int64)
    out = np.empty(len(bins), dtype=np.clongdouble)
    for i, k in enumerate(bins):
        m = (j * int(k)) % n
        ang = (LD(2) * PI) * m.astype(LD) / LD(n)
        out[i] = np.sum(x * (np.cos(ang) + (1j * sign) * np.sin(ang)))
    return out


def cos_frac(num, den):
    """cos(pi num / den) with the integer angle reduced mod 2 den first"""
    m = np.asarray(num, dtype=np.int64) % (2 * den)
    return np.cos(PI * m.astype(LD) / LD(den))


def sin_frac(num, den):
    m = np.asarray(num, dtype=np.int64) % (2 * den)
    return np.sin(PI * m.astype(LD) / LD(den))


def r2r_ld(x, kind):
    """the eleven r2r kinds by their defining sums (reference.texi:2058-2230), long double.
    kind = FFTW's enum value: R2HC 0, HC2R 1, DHT 2, REDFT00 3, REDFT01 4, REDFT10 5, REDFT11 6,
    RODFT00 7, RODFT01 8, RODFT10 9, RODFT11 10"""
    x = np.asarray(x, dtype=LD)
    n = x.shape[0]
    j = np.arange(n, dtype=np.int64)
    k = np.arange(n, dtype=np.int64)[:, None]
    if kind == 0:        # R2HC: r0 r1 ... r(n/2) i((n+1)/2-1) ... i1
        X = dft_ld(x.astype(np.clongdouble), -1)
        y = np.empty(n, dtype=LD)
        for q in range(n // 2 + 1):
            y[q] = X[q].real
        for q in range(1, (n + 1) // 2):
            y[n - q] = X[q].imag
        return y
    if kind == 1:        # HC2R: unnormalised inverse of R2HC
        X = np.zeros(n, dtype=np.clongdouble)
        for q in range(n // 2 + 1):
            im = x[n - q] if 0 < q < (n + 1) // 2 else LD(0)
            X[q] = x[q] + 1j * im
        for q in range(n // 2 + 1, n):
            X[q] = np.conj(X[n - q])
        return dft_ld(X, +1).real
    if kind == 2:        # DHT: sum x_j (cos + sin)(2 pi j k / n)
        return np.sum(x[None, :] * (cos_frac(2 * j[None, :] * k, n) + sin_frac(2 * j[None, :] * k, n)), axis=1)
    if kind == 3:        # REDFT00, logical N = 2(n-1)
        sgn = np.where((k[:, 0] % 2) == 0, LD(1), LD(-1))
        mid = 2 * np.sum(x[None, 1:n - 1] * cos_frac(j[None, 1:n - 1] * k, n - 1), axis=1) if n > 2 else LD(0)
        return x[0] + sgn * x[n - 1] + mid
    if kind == 5:        # REDFT10: 2 sum x_j cos(pi (j + 1/2) k / n)
        return 2 * np.sum(x[None, :] * cos_frac((2 * j[None, :] + 1) * k, 2 * n), axis=1)
    if kind == 4:        # REDFT01: x_0 + 2 sum_{j>=1} x_j cos(pi j (k + 1/2) / n)
        return x[0] + 2 * np.sum(x[None, 1:] * cos_frac(j[None, 1:] * (2 * k + 1), 2 * n), axis=1)
    if kind == 6:        # REDFT11: 2 sum x_j cos(pi (j + 1/2)(k + 1/2) / n)
        return 2 * np.sum(x[None, :] * cos_frac((2 * j[None, :] + 1) * (2 * k + 1), 4 * n), axis=1)
    if kind == 7:        # RODFT00, logical N = 2(n+1)
        return 2 * np.sum(x[None, :] * sin_frac((j[None, :] + 1) * (k + 1), n + 1), axis=1)
    if kind == 9:        # RODFT10: 2 sum x_j sin(pi (j + 1/2)(k + 1) / n)
        return 2 * np.sum(x[None, :] * sin_frac((2 * j[None, :] + 1) * (k + 1), 2 * n), axis=1)
    if kind == 8:        # RODFT01: (-1)^k x_{n-1} + 2 sum_{j<n-1} x_j sin(pi (j + 1)(k + 1/2) / n)
        sgn = np.where((k[:, 0] % 2) == 0, LD(1), LD(-1))
        return sgn * x[n - 1] + 2 * np.sum(x[None, :n - 1] * sin_frac((j[None, :n - 1] + 1) * (2 * k + 1), 2 * n), axis=1)
    if kind == 10:       # RODFT11
        return 2 * np.sum(x[None, :] * sin_frac((2 * j[None, :] + 1) * (2 * k + 1), 4 * n), axis=1)
    raise ValueError(kind)


def main():
    g = Drand48(1)
    z = {}
    pick = np.random.default_rng(3)
    # long c2c sizes, sampled bins: Rader (p - 1 smooth), Bluestein above the one-kernel limit, 13-smooth mixed
    for n in (65537, 12289, 8191, 60060):
        x = g.crand(n)
        bins = np.unique(np.concatenate([[0, 1, 2, n // 2, n - 2, n - 1], pick.integers(0, n, 90)])).astype(np.int64)
        z["c%d_in" % n] = x
        z["c%d_bins" % n] = bins
        z["c%d_fwd" % n] = dft_bins_ld(x, bins, -1).astype(np.complex128)
        z["c%d_bwd" % n] = dft_bins_ld(x, bins, +1).astype(np.complex128)
    # odd-length real transforms (rdft2 via the full complex DFT in the product; rdft2_rdft in the reference)
    for n in (77, 1001):
        x = g.rrand(n)
        z["r%d_in" % n] = x
        z["r%d_out" % n] = dft_ld(x, -1)[: n // 2 + 1].astype(np.complex128)
    # 3-D
    for shape in ((6, 10, 8), (5, 6, 7)):
        key = "x".join(str(s) for s in shape)
        x = g.crand(int(np.prod(shape))).reshape(shape)
        y = dft_ld(dft_ld(dft_ld(x, -1, axis=2), -1, axis=1), -1, axis=0)
        z["c3_%s_in" % key] = x
        z["c3_%s_fwd" % key] = y.astype(np.complex128)
        xr = g.rrand(int(np.prod(shape))).reshape(shape)
        yr = dft_ld(dft_ld(dft_ld(xr, -1, axis=2), -1, axis=1), -1, axis=0)
        z["r3_%s_in" % key] = xr
        z["r3_%s_out" % key] = yr[:, :, : shape[2] // 2 + 1].astype(np.complex128)
    # every r2r kind, one even and one odd length
    for n in (16, 15, 1000, 243):
        for kind in range(11):
            x = g.rrand(n)
            z["k%d_n%d_in" % (kind, n)] = x
            z["k%d_n%d_out" % (kind, n)] = r2r_ld(x, kind).astype(np.float64)
    np.savez(os.path.join(HERE, "pins2.npz"), **z)
    print("wrote pins2.npz:", len(z), "arrays")


if __name__ == "__main__":
    main()
