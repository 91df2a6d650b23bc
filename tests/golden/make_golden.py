"""Generates the golden fixtures in this directory.

The reference ships no stored answers (its tests are self-checking, SURVEY.md
section 4) and cannot be built under this project's rules, so the fixtures pin
the mathematical definition instead: inputs are the reference verifier's
pseudo-random sequence -- drand48() - 0.5 after srand48(1)
(fftw/libbench2/verify-lib.c:64-67, bench-main.c:74), reproduced here from the
documented 48-bit LCG -- and expected outputs are the DFT evaluated directly in
80-bit long double (64-bit mantissa) with exactly reduced angles, then rounded
to double.  Against these, double-precision FFTW itself scores ~1e-16..1e-15.

Run:  python tests/golden/make_golden.py      (rewrites *.npz next to it)
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Drand48(object):
    """POSIX drand48: X' = (0x5DEECE66D X + 0xB) mod 2^48; srand48(s): X = s<<16 | 0x330E"""

    def __init__(self, seed=1):
        self.x = ((seed & 0xFFFFFFFF) << 16) | 0x330E

    def next(self):
        self.x = (0x5DEECE66D * self.x + 0xB) & ((1 << 48) - 1)
        return self.x / float(1 << 48)

    def crand(self, n):
        """n complex values, re then im per element, each drand48() - 0.5 (arand order)"""
        out = np.empty(n, dtype=np.complex128)
        for i in range(n):
            re = self.next() - 0.5
            im = self.next() - 0.5
            out[i] = re + 1j * im
        return out

    def rrand(self, n):
        return np.array([self.next() - 0.5 for _ in range(n)])


def dft_matrix_rows(n, k0, k1, sign):
    """rows k0..k1 of exp(sign 2 pi i j k / n) in long double, angles reduced exactly"""
    k = np.arange(k0, k1, dtype=np.int64)[:, None]
    j = np.arange(n, dtype=np.int64)[None, :]
    m = (k * j) % n
    ang = (2 * np.pi * np.longdouble(1)) * m.astype(np.longdouble) / np.longdouble(n)
    # 2*pi in long double
    two_pi = np.longdouble(2) * np.arctan2(np.longdouble(0), np.longdouble(-1))
    ang = two_pi * m.astype(np.longdouble) / np.longdouble(n)
    return np.cos(ang) + (1j * sign) * np.sin(ang)


def dft_ld(x, sign=-1, axis=-1):
    """direct DFT along `axis` in long double"""
    x = np.moveaxis(np.asarray(x, dtype=np.clongdouble), axis, -1)
    n = x.shape[-1]
    out = np.empty(x.shape, dtype=np.clongdouble)
    step = max(1, (1 << 22) // max(1, n))
    for k0 in range(0, n, step):
        k1 = min(n, k0 + step)
        W = dft_matrix_rows(n, k0, k1, sign)            # (rows, n)
        out[..., k0:k1] = np.tensordot(x, W, axes=([-1], [1]))
    return np.moveaxis(out, -1, axis)


def main():
    g = Drand48(1)
    c2c = {}
    for n in [1, 2, 3, 4, 5, 7, 8, 11, 13, 16, 17, 25, 31, 32, 61, 64, 77, 97, 143,
              1009, 1024, 1031, 4096, 15015]:
        x = g.crand(n)
        c2c["n%d_in" % n] = x
        c2c["n%d_fwd" % n] = dft_ld(x, -1).astype(np.complex128)
        c2c["n%d_bwd" % n] = dft_ld(x, +1).astype(np.complex128)
    # three vectors per size for the batched / interleaved layouts (N*V and NvV)
    for n in [4, 13, 64, 1024]:
        x = g.crand(3 * n).reshape(3, n)
        c2c["v3n%d_in" % n] = x
        c2c["v3n%d_fwd" % n] = dft_ld(x, -1).astype(np.complex128)
    np.savez(os.path.join(HERE, "c2c_1d.npz"), **c2c)

    r2c = {}
    for n in [2, 3, 4, 8, 15, 16, 64, 128, 4096]:
        x = g.rrand(n)
        full = dft_ld(x, -1)
        r2c["n%d_in" % n] = x
        r2c["n%d_out" % n] = full[: n // 2 + 1].astype(np.complex128)
    np.savez(os.path.join(HERE, "r2c_1d.npz"), **r2c)

    nd = {}
    for shape in [(4, 4), (8, 16), (16, 8), (13, 11), (64, 64)]:
        key = "x".join(str(s) for s in shape)
        x = g.crand(shape[0] * shape[1]).reshape(shape)
        y = dft_ld(dft_ld(x, -1, axis=1), -1, axis=0)
        nd["c%s_in" % key] = x
        nd["c%s_fwd" % key] = y.astype(np.complex128)
        xr = g.rrand(shape[0] * shape[1]).reshape(shape)
        yr = dft_ld(dft_ld(xr, -1, axis=1), -1, axis=0)
        nd["r%s_in" % key] = xr
        nd["r%s_out" % key] = yr[:, : shape[1] // 2 + 1].astype(np.complex128)
    np.savez(os.path.join(HERE, "nd.npz"), **nd)
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
