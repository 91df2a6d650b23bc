"""Shared test helpers: oracle bindings, the reference's error metric, RNG."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE = None

# tolerance of the reference's own verifier for double precision
# (reference fftw/libbench2/bench-main.c:70)
TOL = 1e-10


def oracle():
    global _ORACLE
    if _ORACLE is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        lib = C.CDLL(path)
        ip, vp = C.POINTER(C.c_int), C.c_void_p
        lib.oracle_dft_many.argtypes = [C.c_int, ip, C.c_int, vp, ip, C.c_int, C.c_int,
                                        vp, ip, C.c_int, C.c_int, C.c_int]
        lib.oracle_r2c_many.argtypes = [C.c_int, ip, C.c_int, vp, ip, C.c_int, C.c_int,
                                        vp, ip, C.c_int, C.c_int]
        lib.oracle_c2r_many.argtypes = [C.c_int, ip, C.c_int, vp, ip, C.c_int, C.c_int,
                                        vp, ip, C.c_int, C.c_int]
        lib.oracle_r2r_many.argtypes = [C.c_int, ip, C.c_int, vp, ip, C.c_int, C.c_int,
                                        vp, ip, C.c_int, C.c_int, ip]
        lib.oracle_r2r_many.restype = C.c_int
        lib.oracle_r2r_direct.argtypes = [C.c_int, C.c_int, vp, vp]
        lib.oracle_r2r_direct.restype = C.c_int
        lib.oracle_cexp.argtypes = [C.c_longlong, C.c_longlong, C.POINTER(C.c_double)]
        for f in (lib.oracle_dft_many, lib.oracle_r2c_many, lib.oracle_c2r_many):
            f.restype = C.c_int
        _ORACLE = lib
    return _ORACLE


def _ints(v):
    return None if v is None else (C.c_int * len(v))(*v)


def oracle_dft(x, shape, howmany=1, sign=-1, inembed=None, istride=1, idist=None,
               onembed=None, ostride=1, odist=None, out=None):
    """fftw_plan_many_dft + execute through the oracle; x, out: complex128 arrays"""
    n = int(np.prod(shape))
    idist = n if idist is None else idist
    odist = n if odist is None else odist
    if out is None:
        out = np.zeros(max(1, howmany) * n, dtype=np.complex128)
    x = np.ascontiguousarray(x)
    rc = oracle().oracle_dft_many(len(shape), _ints(list(shape)), howmany, x.ctypes.data,
                                  _ints(inembed), istride, idist, out.ctypes.data,
                                  _ints(onembed), ostride, odist, sign)
    assert rc == 0
    return out


def oracle_r2c(x, shape, howmany=1, out=None, inembed=None, istride=1, idist=None,
               onembed=None, ostride=1, odist=None):
    n = int(np.prod(shape))
    hn = n // shape[-1] * (shape[-1] // 2 + 1)
    idist = n if idist is None else idist
    odist = hn if odist is None else odist
    if out is None:
        out = np.zeros(max(1, howmany) * hn, dtype=np.complex128)
    x = np.ascontiguousarray(x)
    rc = oracle().oracle_r2c_many(len(shape), _ints(list(shape)), howmany, x.ctypes.data,
                                  _ints(inembed), istride, idist, out.ctypes.data,
                                  _ints(onembed), ostride, odist)
    assert rc == 0
    return out


def oracle_c2r(y, shape, howmany=1, out=None, inembed=None, istride=1, idist=None,
               onembed=None, ostride=1, odist=None):
    n = int(np.prod(shape))
    hn = n // shape[-1] * (shape[-1] // 2 + 1)
    idist = hn if idist is None else idist
    odist = n if odist is None else odist
    if out is None:
        out = np.zeros(max(1, howmany) * n, dtype=np.float64)
    y = np.ascontiguousarray(y)
    rc = oracle().oracle_c2r_many(len(shape), _ints(list(shape)), howmany, y.ctypes.data,
                                  _ints(inembed), istride, idist, out.ctypes.data,
                                  _ints(onembed), ostride, odist)
    assert rc == 0
    return out


def oracle_r2r(x, shape, kinds, howmany=1, out=None, inembed=None, istride=1, idist=None,
               onembed=None, ostride=1, odist=None):
    """fftw_plan_many_r2r + execute through the oracle; x, out: float64 arrays"""
    n = int(np.prod(shape)) if len(shape) else 1
    idist = n if idist is None else idist
    odist = n if odist is None else odist
    if out is None:
        out = np.zeros(max(1, howmany) * n, dtype=np.float64)
    x = np.ascontiguousarray(x)
    rc = oracle().oracle_r2r_many(len(shape), _ints(list(shape)), howmany, x.ctypes.data,
                                  _ints(inembed), istride, idist, out.ctypes.data,
                                  _ints(onembed), ostride, odist, _ints(list(kinds)))
    assert rc == 0
    return out


def oracle_r2r_direct(x, kind):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    rc = oracle().oracle_r2r_direct(kind, x.size, x.ctypes.data, y.ctypes.data)
    assert rc == 0
    return y


def aerror(a, b):
    """Relative L-infinity error of the reference's verifier
    (reference fftw/libbench2/verify-lib.c:36-58): max |a-b| over
    max min(|a|,|b|), with |z| = max(|re|, |im|)."""
    a = np.asarray(a).reshape(-1)
    b = np.asarray(b).reshape(-1)
    if a.size == 0:
        return 0.0
    a = a.astype(np.complex128)
    b = b.astype(np.complex128)
    d = a - b
    e = np.maximum(np.abs(d.real), np.abs(d.imag)).max()
    na = np.maximum(np.abs(a.real), np.abs(a.imag))
    nb = np.maximum(np.abs(b.real), np.abs(b.imag))
    mag = np.minimum(na, nb).max()
    if mag == 0.0:
        return 0.0 if e == 0.0 else np.inf
    assert not np.isnan(e)
    return float(e / mag)


def crand(rng, *shape):
    """uniform [-0.5, 0.5) real and imaginary parts, like the reference's
    mydrand() = drand48() - 0.5 (verify-lib.c:64-67)"""
    return (rng.random(shape) - 0.5) + 1j * (rng.random(shape) - 0.5)


def rrand(rng, *shape):
    return rng.random(shape) - 0.5
