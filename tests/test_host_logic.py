"""CPU-tier tests of the host side: the planner's step lists are interpreted
with numpy (tests/step_interp.py) and compared with the oracle; the API layer's
argument handling follows the reference (fftw/fftw_api.c:560-880); the shared
library exports every symbol the headers declare.  No transform is executed by
the library here: it has no CPU path."""
import os
import re
import subprocess

import numpy as np
import pytest

import fftw3_amd as fa
from step_interp import run_plan_on_host
from util import ROOT, TOL, aerror, crand, oracle_c2r, oracle_dft, oracle_r2c, rrand

rng = np.random.default_rng(11)


def _c2c(n, b, sign, inplace=False):
    x = crand(rng, b, n)
    x0 = x.copy()
    y = x if inplace else np.zeros_like(x)
    p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, sign)
    run_plan_on_host(p, x, y)
    return aerror(y, oracle_dft(x0, (n,), b, sign).reshape(b, n)), p


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 13, 16, 17, 25, 31, 32, 61, 64, 77,
                               97, 100, 143, 1009, 1024, 1031, 4096, 5000, 8192, 15015, 17408, 65536])
def test_planner_c2c_1d(n):
    for b in (1, 3):
        for sign in (-1, 1):
            e, _ = _c2c(n, b, sign)
            assert e < TOL
    e, _ = _c2c(n, 2, -1, inplace=True)
    assert e < TOL


def test_planner_decompositions():
    """what the static planner does for the BASELINE shapes"""
    x = np.zeros(1 << 20, dtype=complex)
    p = fa.plan_dft_1d(1 << 20, x, x.copy(), fa.FORWARD)
    s = p.steps()
    assert [d.L for d in s] == [1024, 1024]
    # two passes: the inter-pass twiddle rides on the input of the second one
    assert s[0].tw_n == 0 and s[1].tw_n == 1 << 20 and (s[1].flags & fa.F_TW_IN)
    assert [d.variant for d in s] == [1, 1]          # both run the register-resident kernel
    assert fa.factor_passes(3 * 5 * 7 * 11 * 13 * 1024) and \
        int(np.prod(fa.factor_passes(3 * 5 * 7 * 11 * 13 * 1024))) == 15375360
    assert fa.factor_passes(1024) == [1024]
    # 97 -> Rader (p-1 = 96 smooth), 1031 -> Bluestein, both visible in the plan print
    xs = np.zeros(1031, dtype=complex)
    assert "rader-mul" in fa.plan_dft_1d(97, xs, xs.copy(), fa.FORWARD).sprint()
    assert "copy" in fa.plan_dft_1d(1031, xs, xs.copy(), fa.FORWARD).sprint()


@pytest.mark.parametrize("shape", [(4, 4), (8, 16), (16, 8), (13, 11), (64, 64), (3, 5, 7),
                                   (2, 1024), (600, 6), (1030, 4), (5, 2048), (2, 3, 4, 5)])
def test_planner_c2c_nd(shape):
    nn = int(np.prod(shape))
    for b in (1, 2):
        for sign in (-1, 1):
            x = crand(rng, b, *shape)
            y = np.zeros_like(x)
            p = fa.plan_many_dft(len(shape), list(shape), b, x, None, 1, nn, y, None, 1, nn, sign)
            run_plan_on_host(p, x, y)
            assert aerror(y, oracle_dft(x, shape, b, sign).reshape(x.shape)) < TOL


def test_planner_strides_embed_and_chunks():
    # interleaved vectors (NvV), single and multi pass
    for n, b in ((48, 5), (6000, 3)):
        x = crand(rng, n, b)
        y = np.zeros_like(x)
        p = fa.plan_many_dft(1, [n], b, x, None, b, 1, y, None, b, 1, fa.FORWARD)
        run_plan_on_host(p, x, y)
        assert aerror(y, oracle_dft(x, (n,), b, istride=b, idist=1, ostride=b, odist=1)
                      .reshape(n, b)) < TOL
    # sub-array of a larger 2-D array through inembed/onembed
    x = crand(rng, 12, 20)
    y = np.zeros((10, 16), dtype=complex)
    p = fa.plan_many_dft(2, [6, 8], 1, x, [12, 20], 1, 0, y, [10, 16], 1, 0, fa.FORWARD)
    run_plan_on_host(p, x, y)
    assert aerror(y[:6, :8], np.fft.fft2(x[:6, :8])) < TOL
    assert np.all(y[6:] == 0) and np.all(y[:, 8:] == 0)
    # batch larger than one chunk: every chunk must land in its own slice
    fa.set_chunk_bytes(1 << 16)
    try:
        n, b = 8192, 7
        x = crand(rng, b, n)
        y = np.zeros_like(x)
        p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, fa.FORWARD)
        assert p.chunk < b
        run_plan_on_host(p, x, y)
        assert aerror(y, oracle_dft(x, (n,), b).reshape(b, n)) < TOL
    finally:
        fa.set_chunk_bytes(0)


@pytest.mark.parametrize("n", [2, 3, 4, 8, 15, 16, 64, 128, 4096, 10000, 17, 97, 1009, 2018, 32768])
def test_planner_r2c_c2r_1d(n):
    for b in (1, 3):
        x = rrand(rng, b, n)
        y = np.zeros((b, n // 2 + 1), dtype=complex)
        p = fa.plan_many_dft_r2c(1, [n], b, x, None, 1, n, y, None, 1, n // 2 + 1)
        run_plan_on_host(p, x, y)
        ref = oracle_r2c(x, (n,), b).reshape(b, n // 2 + 1)
        assert aerror(y, ref) < TOL
        assert np.all(y[:, 0].imag == 0)
        yy = ref.copy()
        yy[:, 0] += 0.25j                      # Im Y[0] must be ignored by c2r
        z = np.zeros((b, n))
        p = fa.plan_many_dft_c2r(1, [n], b, yy, None, 1, n // 2 + 1, z, None, 1, n)
        run_plan_on_host(p, yy, z)
        assert aerror(z, oracle_c2r(ref, (n,), b).reshape(b, n)) < TOL
        assert np.array_equal(yy[:, 1:], ref[:, 1:])     # input preserved


def test_planner_r2c_inplace_padded_and_nd():
    n, b = 64, 3
    buf = np.zeros((b, 2 * (n // 2 + 1)))
    x = rrand(rng, b, n)
    buf[:, :n] = x
    p = fa.plan_many_dft_r2c(1, [n], b, buf, None, 1, 2 * (n // 2 + 1), buf, None, 1, n // 2 + 1)
    run_plan_on_host(p, buf, buf)
    assert aerror(buf.view(complex), oracle_r2c(x, (n,), b).reshape(b, n // 2 + 1)) < TOL
    for shape in [(4, 4), (8, 16), (13, 11), (16, 9), (64, 64), (5, 6, 8)]:
        nn = int(np.prod(shape))
        hs = shape[:-1] + (shape[-1] // 2 + 1,)
        hh = int(np.prod(hs))
        x = rrand(rng, 2, *shape)
        y = np.zeros((2,) + hs, dtype=complex)
        p = fa.plan_many_dft_r2c(len(shape), list(shape), 2, x, None, 1, nn, y, None, 1, hh)
        run_plan_on_host(p, x, y)
        ref = oracle_r2c(x, shape, 2).reshape(y.shape)
        assert aerror(y, ref) < TOL
        z = np.zeros_like(x)
        yy = ref.copy()
        p = fa.plan_many_dft_c2r(len(shape), list(shape), 2, yy, None, 1, hh, z, None, 1, nn)
        run_plan_on_host(p, yy, z)
        assert aerror(z, x * nn) < TOL


def test_planner_radix4_real_transforms(monkeypatch):
    """n = 4m: two quarter-length complex DFTs + radix-4 untangle (the reference's
    rdft2-ct-dit/4 + hc2cfdft_4 plan); chosen by itself when m
    needs fewer passes than n/2 (n = 2^22: 2^20 = 1024 x 1024 against three passes for 2^21), forced for the rest;
    a half length with a one-trip rows kernel (2048, 5000 ...) is never beaten"""
    assert "untangle4" not in fa.plan_dft_r2c_1d(4096, np.zeros(4096), np.zeros(2049, dtype=complex)).sprint()
    assert "untangle4" not in fa.plan_many_dft_r2c(1, [10000], 8, np.zeros(8), None, 1, 10000, np.zeros(8, dtype=complex), None,
                                                   1, 5001).sprint()
    assert "untangle4" in fa.plan_dft_r2c_1d(1 << 22, np.zeros(8), np.zeros(8, dtype=complex)).sprint()
    monkeypatch.setenv("FFTW_AMD_FORCE_RADIX4", "1")
    for n in (8, 12, 20, 36, 64, 100, 1000, 4096, 40000):
        for b in (1, 3):
            x = rrand(rng, b, n)
            y = np.zeros((b, n // 2 + 1), dtype=complex)
            p = fa.plan_many_dft_r2c(1, [n], b, x, None, 1, n, y, None, 1, n // 2 + 1)
            assert "untangle4" in p.sprint()
            run_plan_on_host(p, x, y)
            ref = oracle_r2c(x, (n,), b).reshape(b, n // 2 + 1)
            assert aerror(y, ref) < TOL, n
            yy = ref.copy()
            yy[:, 0] += 0.25j
            yy[:, -1] += 0.5j * (n % 2 == 0)
            z = np.zeros((b, n))
            p = fa.plan_many_dft_c2r(1, [n], b, yy, None, 1, n // 2 + 1, z, None, 1, n)
            assert "tangle4" in p.sprint()
            run_plan_on_host(p, yy, z)
            assert aerror(z, oracle_c2r(ref, (n,), b).reshape(b, n)) < TOL, n
    for shape in [(4, 8), (6, 16), (5, 6, 12)]:
        nn, hs = int(np.prod(shape)), shape[:-1] + (shape[-1] // 2 + 1,)
        x = rrand(rng, 2, *shape)
        y = np.zeros((2,) + hs, dtype=complex)
        p = fa.plan_many_dft_r2c(len(shape), list(shape), 2, x, None, 1, nn, y, None, 1, int(np.prod(hs)))
        run_plan_on_host(p, x, y)
        assert aerror(y, oracle_r2c(x, shape, 2).reshape(y.shape)) < TOL


def test_guru_and_split_interfaces():
    # guru: transform dim of 30 with stride 7 (complex), two howmany dims
    n, h0, h1 = 30, 3, 2
    x = crand(rng, h0, h1, n * 7)
    y = np.zeros((h0, h1, n), dtype=complex)
    p = fa.plan_guru64_dft([(n, 7, 1)], [(h0, h1 * n * 7, h1 * n), (h1, n * 7, n)], x, y, fa.FORWARD)
    run_plan_on_host(p, x, y)
    assert aerror(y, np.fft.fft(x[:, :, ::7], axis=2)) < TOL
    # split arrays: separate real and imaginary planes
    n = 96
    re, im = rrand(rng, 2, n)
    planes = np.stack([re, im])
    outp = np.zeros_like(planes)
    p = fa.plan_guru64_split_dft([(n, 1, 1)], [], planes[0], planes[1], outp[0], outp[1])
    # interpret on the flat buffers: in = planes, out = outp
    from step_interp import Interp, scratch_reals
    Interp(p).run(planes.reshape(-1), outp.reshape(-1), scratch_reals(p))
    assert aerror(outp[0] + 1j * outp[1], np.fft.fft(re + 1j * im)) < TOL


def test_api_argument_handling():
    """NULL for what the reference rejects (fftw_many_kosherp, A.c:863-877;
    in-place location check, A.c:4090-4094)"""
    x = np.zeros(64, dtype=complex)
    y = np.zeros(64, dtype=complex)
    with pytest.raises(ValueError):
        fa.plan_dft_1d(0, x, y, fa.FORWARD)                  # n must be > 0
    with pytest.raises(ValueError):
        fa.plan_many_dft(1, [8], -1, x, None, 1, 8, y, None, 1, 8, fa.FORWARD)   # howmany < 0
    with pytest.raises(ValueError):
        fa.plan_many_dft(1, [8], 2, x, None, 1, 8, x, None, 2, 16, fa.FORWARD)   # in place, strides differ
    p = fa.plan_many_dft(1, [8], 0, x, None, 1, 8, y, None, 1, 8, fa.FORWARD)    # howmany == 0 is legal
    assert p.batch == 0 and len(p.steps()) == 0
    p = fa.plan_dft(0, [], x, y, fa.FORWARD)                 # rank 0 = copy
    assert len(p.steps()) == 1 and p.steps()[0].L == 1
    import ctypes as C
    assert fa.lib.fftw_plan_r2r_1d(8, None, None, 0, 0) is None or True
    fa.lib.fftw_destroy_plan(None)                           # NULL-safe (A.c:409-410)
    a, m, f = fa.plan_dft_1d(1024, x.repeat(16), x.repeat(16).copy(), fa.FORWARD).flops()
    assert abs((a + m + 2 * f) - 5 * 1024 * 10) / (5 * 1024 * 10) < 0.5


def test_wisdom_round_trip_and_wisdom_only(tmp_path):
    """wisdom is text that survives export -> forget -> import; FFTW_WISDOM_ONLY plans
    only problems it covers; a malformed record leaves the old wisdom intact
    (reference: transactional import, fftw/fftw_api.c:15577-15581)"""
    fa.forget_wisdom()
    x = np.zeros(1 << 14, dtype=complex)
    y = x.copy()
    with pytest.raises(ValueError):
        fa.plan_dft_1d(1 << 14, x, y, fa.FORWARD, fa.ESTIMATE | fa.WISDOM_ONLY)
    key = "t0 s-1 r1 16384:2:2 h0 i1 o1 p0"
    rec = "(fftw3_amd_wisdom-1\n  (%s) 268435456 0 512 1 0.125000\n)\n" % key
    assert fa.import_wisdom_from_string(rec) == 1
    p = fa.plan_dft_1d(1 << 14, x, y, fa.FORWARD, fa.ESTIMATE | fa.WISDOM_ONLY)
    assert [d.L for d in p.steps()] == [128, 128]
    text = fa.export_wisdom_to_string()
    assert key in text and "268435456 0 512 1" in text
    assert fa.import_wisdom_from_string("(fftw3_amd_wisdom-1\n  (broken 1 2\n)") == 0
    assert fa.import_wisdom_from_string("garbage") == 0
    assert fa.export_wisdom_to_string() == text                # failed imports changed nothing
    f = str(tmp_path / "wisdom.txt")
    assert fa.lib.fftw_export_wisdom_to_filename(f.encode()) == 1
    fa.forget_wisdom()
    assert key not in fa.export_wisdom_to_string()
    assert fa.lib.fftw_import_wisdom_from_filename(f.encode()) == 1
    assert fa.export_wisdom_to_string() == text
    fa.forget_wisdom()


def test_execute_without_device_fails_loudly():
    if fa.device_count() > 0:
        pytest.skip("a HIP device is present")
    x = np.zeros(16, dtype=complex)
    p = fa.plan_dft_1d(16, x, x.copy(), fa.FORWARD)
    with pytest.raises(RuntimeError):
        p.execute()


def _declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(fftw_(?:amd_)?[a-z0-9_]+)\s*\(", txt))
    names |= set(re.findall(r"extern const char (fftw_[a-z_]+)\[\]", txt))
    return {n for n in names if not n.endswith("_func")}


def test_library_exports_every_declared_symbol():
    lib = os.path.join(ROOT, "fftw3_amd", "lib", "libfftw3_amd.so")
    out = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE, text=True,
                         check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    want = _declared_symbols("fftw3.h") | _declared_symbols("fftw3_amd.h")
    assert len(want) >= 85
    missing = sorted(want - exported)
    assert not missing, "declared but not exported: %s" % missing


def test_host_twiddles_match_oracle_bit_for_bit():
    import ctypes as C
    from util import oracle
    out = (C.c_double * 2)()
    for n in (7, 1024, 1 << 20, 15375360):
        for m in (0, 1, 5, n // 8, n // 3, n // 2, n - 1):
            oracle().oracle_cexp(m, n, out)
            assert fa.cexp(m, n) == (out[0], out[1])
    assert fa.lib.fftw_amd_find_generator(97) == 5 and fa.lib.fftw_amd_find_generator(65537) == 3
    assert fa.lib.fftw_amd_power_mod(3, 65536, 65537) == 1


def test_planner_random_composite_shapes():
    """CPU-tier twin of the GPU random sweep (fftw/tests/check.pl:186-251): the planner's
    step lists for random shapes / vector types, interpreted with numpy, against the oracle"""
    r = np.random.default_rng(4321)
    primes = [2, 3, 5, 7, 11, 13]
    for case in range(40):
        rank = int(r.integers(1, 4))
        shape = []
        budget = 20000 ** (1.0 / rank)
        for _ in range(rank):
            n = 1
            while True:
                f = primes[int(r.integers(0, len(primes)))]
                if n * f > max(2.0, budget):
                    break
                n *= f
                if r.random() < 0.25:
                    break
            shape.append(n)
        shape = tuple(shape)
        nn = int(np.prod(shape))
        vtype = int(r.integers(0, 3))
        v = 1 if vtype == 0 else int(r.integers(2, 5))
        sign = -1 if r.random() < 0.5 else 1
        if vtype == 2:
            x = crand(r, *(shape + (v,)))
            istride, idist = v, 1
        else:
            x = crand(r, *((v,) + shape))
            istride, idist = 1, nn
        y = np.zeros_like(x)
        p = fa.plan_many_dft(rank, list(shape), v, x, None, istride, idist, y, None, istride, idist, sign)
        run_plan_on_host(p, x, y)
        ref = oracle_dft(x, shape, v, sign, istride=istride, idist=idist, ostride=istride, odist=idist)
        assert aerror(y.reshape(-1), ref) < TOL, (case, shape, v, vtype, sign, p.sprint())


def test_guru_r2c_column_major_real_input_with_rader_half_length():
    """regression: a non-batch loop of stride 1 (real data) around a Rader sub-transform -- the
    scratch layout must still put the transform axis innermost (the rader-mul step walks
    contiguous vectors)"""
    import fftw3_amd as fa
    from step_interp import run_plan_on_host
    from util import oracle_r2c, rrand
    rng = np.random.default_rng(31)
    n0, n1 = 3, 106                      # half length 53: prime, Rader over 52
    nh = n1 // 2 + 1
    xt = rrand(rng, n1, n0)              # memory [n1][n0]: x[i0][i1] = xt[i1][i0]
    x = xt.reshape(-1).copy()
    y = np.zeros(n0 * nh, dtype=np.complex128)
    p = fa.plan_guru64_dft_r2c([(n0, 1, nh), (n1, n0, 1)], [], x, y)
    assert "rader-mul" in p.sprint()
    run_plan_on_host(p, x, y)
    assert aerror(y, oracle_r2c(np.ascontiguousarray(xt.T).reshape(-1), [n0, n1], 1)) < TOL


def test_planning_from_many_threads():
    """ctypes drops the GIL: planner calls really run concurrently; plans must equal the serial ones"""
    import threading
    import fftw3_amd as fa
    sizes = [64, 1000, 4096, 10007, 1 << 16, 15015, 53 * 2, 360]
    x = np.zeros(1 << 17, dtype=np.complex128)
    serial = {n: fa.plan_dft_1d(n, x, x, fa.FORWARD).sprint() for n in sizes}
    errs = []

    def work(seed):
        rng = np.random.default_rng(seed)
        for _ in range(40):
            n = sizes[int(rng.integers(0, len(sizes)))]
            try:
                if fa.plan_dft_1d(n, x, x, fa.FORWARD).sprint() != serial[n]:
                    errs.append(n)
                fa.import_wisdom_from_string(fa.export_wisdom_to_string())
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))
    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:5]


C_CLIENT = r"""
/* a stock FFTW client: nothing here knows about the GPU library */
#include <stdio.h>
#include <fftw3.h>
int main(void) {
    int n = 1024, ok = 1;
    fftw_complex *in = fftw_alloc_complex(n), *out = fftw_alloc_complex(n);
    double *r = fftw_alloc_real(n);
    fftw_plan p, q, t;
    fftw_r2r_kind k = FFTW_REDFT10;
    if (!in || !out || !r) return 2;
    p = fftw_plan_dft_1d(n, in, out, FFTW_FORWARD, FFTW_ESTIMATE);
    q = fftw_plan_dft_r2c_1d(n, r, out, FFTW_ESTIMATE);
    t = fftw_plan_r2r_1d(n, r, r, k, FFTW_ESTIMATE);
    ok = p && q && t && fftw_plan_dft_1d(0, in, out, FFTW_FORWARD, FFTW_ESTIMATE) == NULL;
    {
        double add, mul, fma;
        fftw_flops(p, &add, &mul, &fma);
        ok = ok && add + mul + 2 * fma > 0;
        fftw_print_plan(t);
        printf("\n");
    }
    fftw_destroy_plan(p); fftw_destroy_plan(q); fftw_destroy_plan(t);
    fftw_free(in); fftw_free(out); fftw_free(r);
    fftw_cleanup();
    printf(ok ? "client ok\n" : "client FAILED\n");
    return ok ? 0 : 1;
}
"""


def test_stock_c_client_compiles_and_links_as_libfftw3(tmp_path):
    """drop-in boundary: a plain FFTW program builds against include/fftw3.h and links with
    -lfftw3 from fftw3_amd/lib (planning only here: no GPU in this tier)"""
    src = tmp_path / "client.c"
    exe = tmp_path / "client"
    src.write_text(C_CLIENT)
    libdir = os.path.join(ROOT, "fftw3_amd", "lib")
    subprocess.run(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                    "-L", libdir, "-lfftw3", "-Wl,-rpath," + libdir, "-lm", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "client ok" in r.stdout and "rdft-r2r" in r.stdout, r.stdout


def test_wisdom_key_of_rank8_plan_with_huge_strides_stays_in_bounds():
    """the wisdom key is built from every dim's (n, is, os): eight dims of 18-digit strides used to
    run past a 320-byte stack buffer (snprintf returns the would-be length) -- "stack smashing
    detected" before any planning.  Keys must also stay distinct when only a late stride differs."""
    big = 10 ** 17
    x = np.zeros(16, dtype=complex)
    y = np.zeros(16, dtype=complex)
    for hm in ([], [(2, 8, 8)]):
        for rank in (7, 8):
            dims = [(1, big + i, big - i) for i in range(rank - 1)] + [(8, 1, 1)]
            try:
                p = fa.plan_guru64_dft(dims, hm, x, y, fa.FORWARD, fa.ESTIMATE)
            except ValueError:
                continue                 # more loop dims than the executor carries: NULL, not a crash
            assert p.handle and p.sprint()
    fa.forget_wisdom()
    dims = [(1, big + i, big - i) for i in range(7)] + [(8, 1, 1)]
    p = fa.plan_guru64_dft(dims, [], x, y, fa.FORWARD, fa.ESTIMATE)
    run_plan_on_host(p, x, y)


def _pass_lengths(p):
    return [s.L for s in p.steps() if s.kind == 1]


def test_split_models_keep_the_factorisation_exact_and_prefer_measured_winners():
    """the measured cost models (split_costs.inc / split2_costs.inc) only reorder and re-factor passes:
    the product of the pass lengths is n, every length has a register kernel; spot checks of picks the
    sweeps established (profiles/r02_*_split_samples.jsonl)"""
    x = np.zeros(4, dtype=complex)
    for n in (1 << 14, 1 << 16, 1 << 17, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 4 * 10 ** 6, 10 ** 4, 10 ** 5, 10 ** 6, 60060, 518400,
              15375360, 10 ** 7, 6 ** 9, 14817600, 1080 * 1024, 2000 * 1000):
        p = fa.plan_many_dft(1, [n], 64, x, None, 1, n, x.copy(), None, 1, n, fa.FORWARD)
        lens = _pass_lengths(p)
        assert int(np.prod([int(v) for v in lens])) == n, (n, lens)
        assert all("lds" not in l for l in p.sprint().splitlines()[1:]), p.sprint()
    lens = lambda n: _pass_lengths(fa.plan_many_dft(1, [n], 64, x, None, 1, n, x.copy(), None, 1, n, fa.FORWARD))
    assert lens(1 << 16) == [128, 512]                    # not 256 x 256: the 256-point kernel is the slow one
    assert lens(1 << 20) == [1024, 1024]
    assert lens(1 << 21) == [2048, 1024]                  # two trips through the narrow-tile 2048-point kernel
    assert lens(1 << 22) == [2048, 2048]                  # round 3: both passes on the 512-item kernels (8 per tile)
    assert lens(1920 * 2048) == [1920, 2048] and len(lens(1 << 23)) == 3
    three = lens(15375360)
    assert len(three) == 3 and three[0] % 8 == 0          # first length on the 128-byte grid
    # a strided axis of 1025 ... 2048 points runs in one trip (the 1080 of a 1080 x 1920 image)
    p = fa.plan_many_dft(2, [1080, 1920], 4, x, None, 1, 1080 * 1920, x.copy(), None, 1, 1080 * 1920, fa.FORWARD)
    assert _pass_lengths(p) == [1920, 1080], p.sprint()


def test_r2c_decimated_over_the_real_data(monkeypatch):
    """cfg.real_dec (FFTW_AMD_REAL_DEC=1, a FFTW_MEASURE candidate): n = L1 x 2048 real points as the complex pass of
    length L1 over the input read as pairs + the rows step with FFTW_AMD_F_REAL_DEC -- two steps whose list computes
    the half spectrum under the numpy interpreter, out of place and in FFTW's padded in-place layout; not planned
    for FFTW_UNALIGNED, for lengths that are no multiple of 2048, or by default"""
    rng = np.random.default_rng(11)
    n, b = 2048 * 256, 3
    xr = rrand(rng, b, n)
    y = np.zeros((b, n // 2 + 1), dtype=complex)
    assert "real-decimated" not in fa.plan_many_dft_r2c(1, [n], b, xr, None, 1, n, y, None, 1, n // 2 + 1).sprint()
    monkeypatch.setenv("FFTW_AMD_REAL_DEC", "1")
    p = fa.plan_many_dft_r2c(1, [n], b, xr, None, 1, n, y, None, 1, n // 2 + 1)
    st = p.steps()
    assert len(st) == 2 and "real-decimated" in p.sprint(), p.sprint()
    assert (st[1].flags & fa.F_REAL_DEC) and (st[1].flags & fa.F_TW_IN) and st[1].dim_n[0] == 256 // 2 + 1 and st[1].tile == 8
    run_plan_on_host(p, xr, y)
    ref = oracle_r2c(xr, (n,), b).reshape(b, n // 2 + 1)
    assert aerror(y, ref) < TOL
    assert np.all(y[:, 0].imag == 0.0) and np.all(y[:, n // 2].imag == 0.0)
    pad = np.zeros((b, n + 2))
    pad[:, :n] = xr
    q = fa.plan_many_dft_r2c(1, [n], b, pad, None, 1, n + 2, pad, None, 1, n // 2 + 1)
    assert "real-decimated" in q.sprint()
    run_plan_on_host(q, pad, pad)
    assert aerror(pad.reshape(b, n // 2 + 1, 2)[..., 0] + 1j * pad.reshape(b, n // 2 + 1, 2)[..., 1], ref) < TOL
    assert "real-decimated" not in fa.plan_many_dft_r2c(1, [n], b, xr, None, 1, n, y, None, 1, n // 2 + 1,
                                                         fa.ESTIMATE | fa.UNALIGNED).sprint()
    m = 2000 * 300
    assert "real-decimated" not in fa.plan_many_dft_r2c(1, [m], b, np.zeros((b, m)), None, 1, m,
                                                         np.zeros((b, m // 2 + 1), dtype=complex), None, 1, m // 2 + 1).sprint()


def test_radix4_real_plans_only_where_their_pair_tiles_have_register_kernels():
    """n = 4m through two quarter-length transforms on interleaved pairs (r2c-untangle4 / c2r-tangle4) only when a
    multi-pass m is a power of two: other lengths would put a pass on the runtime-radix LDS kernel (3 932 160: 7.9 ms
    per 4 GiB against 4.3 for the half-length plan)"""
    for n in (3932160, 3145728, 4096000):
        xr = np.zeros(8)
        y = np.zeros(8, dtype=complex)
        p = fa.plan_many_dft_r2c(1, [n], 64, xr, None, 1, n, y, None, 1, n // 2 + 1, fa.ESTIMATE)
        assert "lds:" not in p.sprint() and "untangle4" not in p.sprint(), p.sprint()
        q = fa.plan_many_dft_c2r(1, [n], 64, y, None, 1, n // 2 + 1, xr, None, 1, n, fa.ESTIMATE)
        assert "lds:" not in q.sprint(), q.sprint()
    xr = np.zeros(8)
    y = np.zeros(8, dtype=complex)
    assert "untangle4" in fa.plan_many_dft_r2c(1, [1 << 22], 64, xr, None, 1, 1 << 22, y, None, 1, (1 << 21) + 1, fa.ESTIMATE).sprint()


def test_streaming_access_flags_only_on_the_callers_side_and_only_for_large_batches():
    """FFTW_AMD_F_NT_IN / NT_OUT (mark_streaming_accesses): set on the step that reads the caller's input once
    and on the step that writes the output nobody reads back, never on scratch traffic, and not at all when the
    whole batch is small enough to stay cached"""
    NT_IN, NT_OUT = 1 << 12, 1 << 13
    x = np.zeros(4, dtype=complex)
    n = 1 << 20
    small = fa.plan_many_dft(1, [n], 4, x, None, 1, n, x.copy(), None, 1, n, fa.FORWARD)       # 128 MiB touched
    assert all(not (s.flags & (NT_IN | NT_OUT)) for s in small.steps())
    big = fa.plan_many_dft(1, [n], 64, x, None, 1, n, x.copy(), None, 1, n, fa.FORWARD)        # 2 GiB touched
    st = big.steps()
    assert st[0].src_buf == 0 and (st[0].flags & NT_IN) and not (st[0].flags & NT_OUT)         # writes scratch
    assert st[-1].dst_buf == 1 and (st[-1].flags & NT_OUT) and not (st[-1].flags & NT_IN)      # reads scratch
    # 2-D: the first axis writes the output array that the second axis reads back: no NT_OUT there
    p2 = fa.plan_many_dft(2, [1024, 2048], 64, x, None, 1, 1 << 21, x.copy(), None, 1, 1 << 21, fa.FORWARD)
    s2 = p2.steps()
    assert s2[0].dst_buf == 1 and not (s2[0].flags & NT_OUT)
    assert s2[-1].dst_buf == 1 and (s2[-1].flags & NT_OUT)
    # 4096 x 4096 (round 3: two trips through scratch): NT_IN on the strided pass, NT_OUT on the four-row pass only
    p4 = fa.plan_many_dft(2, [4096, 4096], 8, x, None, 1, 1 << 24, x.copy(), None, 1, 1 << 24, fa.FORWARD)
    s4 = p4.steps()
    assert len(s4) == 2 and s4[0].src_buf == 0 and s4[0].dst_buf >= 2 and (s4[0].flags & NT_IN) and not (s4[0].flags & NT_OUT)
    assert s4[1].src_buf >= 2 and s4[1].dst_buf == 1 and (s4[1].flags & NT_OUT) and not (s4[1].flags & NT_IN)
    assert (s4[1].flags & fa.F_LO_DFT) and s4[1].tile_lo_n == 4


def test_one_trip_rows_plans():
    """the steps without an LDS-kernel fallback (rows above 4096 points, fused real rows above n = 2048 and for mixed
    lengths, the one-kernel Bluestein, the one-stage rows kernel) are planned only where their layout is settled at
    plan time; their step lists compute the right answer under the numpy interpreter"""
    x = np.zeros(8, dtype=complex)
    one = lambda n, b, flags=fa.ESTIMATE: fa.plan_many_dft(1, [n], b, x, None, 1, n, x.copy(), None, 1, n, fa.FORWARD, flags)
    for n in (5000, 6561, 8192, 16384):
        assert len(one(n, 16).steps()) == 1 and "reg3" in one(n, 16).sprint(), n
        assert len(one(n, 16, fa.ESTIMATE | fa.UNALIGNED).steps()) == 2, n      # any alignment: keep the fallback
        assert len(one(n, 1).steps()) == 2, n                                  # nothing to tile over
    assert "pass-16/reg1" in one(16, 4096).sprint() and "reg1" not in one(16, 100).sprint()
    assert "pass-2145/bluestein-rows n=1031" in one(1031, 8).sprint()
    assert "bluestein-rows" not in one(1031, 8, fa.ESTIMATE | fa.UNALIGNED).sprint()
    assert "bluestein-rows" not in one(37, 64).sprint()                        # the ladder would pad 7 x
    # padded lengths above 8192: one row per workgroup of 512 work-items (kernels_bluew.hip), every prime below 8192
    assert "pass-16384/bluestein-rows n=8191" in one(8191, 4).sprint() and len(one(8191, 4).steps()) == 1
    assert "pass-8640/bluestein-rows n=4099" in one(4099, 4).sprint()
    assert "bluestein-rows" not in one(8209, 4).sprint()
    for n, b in ((5000, 3), (16384, 2), (1031, 5), (23, 300), (1018, 2), (4099, 2), (8191, 1)):
        for sign in (-1, 1):
            e, p = _c2c(n, b, sign)
            assert e < TOL, (n, p.sprint())
    e, p = _c2c(16384, 2, -1, inplace=True)
    assert e < TOL and len(p.steps()) == 1
    for n in (1000, 4000, 10000, 16384, 32768):
        b = 3
        xr = rrand(rng, b, n)
        y = np.zeros((b, n // 2 + 1), dtype=complex)
        p = fa.plan_many_dft_r2c(1, [n], b, xr, None, 1, n, y, None, 1, n // 2 + 1)
        assert "r2c-rows" in p.sprint() and len(p.steps()) == 1, p.sprint()
        run_plan_on_host(p, xr, y)
        ref = oracle_r2c(xr, (n,), b).reshape(b, n // 2 + 1)
        assert aerror(y, ref) < TOL, n
        z = np.zeros((b, n))
        q = fa.plan_many_dft_c2r(1, [n], b, ref.copy(), None, 1, n // 2 + 1, z, None, 1, n)
        assert "c2r-rows" in q.sprint() and len(q.steps()) == 1, q.sprint()
        yy = ref.copy()
        run_plan_on_host(q, yy, z)
        assert aerror(z, xr * n) < TOL, n


def test_inplace_problems_with_different_strides_on_the_two_sides():
    """the reference accepts an in-place problem whose input and output strides differ as long as both address the
    same locations (fftw_tensor_inplace_locations, fftw/fftw_api.c:17298-17311; its planner then uses the in-place
    square DIF + transpose codelets "q1_r" or buffers, :2204-2250, :2596-2727): transposed-output transforms and
    rank-0 transposes.  Here such a plan runs through a dense scratch image in one chunk; step lists under the
    numpy interpreter."""
    rng2 = np.random.default_rng(17)
    for n, v in ((64, 64), (48, 80), (7, 5), (1024, 16), (4096, 3)):
        x = crand(rng2, 1, n * v).reshape(-1)
        x0 = x.copy()
        p = fa.plan_guru64_dft([(n, 1, v)], [(v, n, 1)], x, x, fa.FORWARD)      # in[b n + j] -> out[k v + b]
        assert p.batch == p.chunk                                              # one chunk: reads before writes
        run_plan_on_host(p, x, x)
        want = oracle_dft(x0.reshape(1, -1), (n,), v).reshape(v, n).T.reshape(-1)
        assert aerror(x, want) < TOL, (n, v)
    # rank 0: an in-place transpose, non-square
    n0, n1 = 12, 20
    x = crand(rng2, 1, n0 * n1).reshape(-1)
    x0 = x.copy()
    p = fa.plan_guru64_dft([], [(n0, n1, 1), (n1, 1, n0)], x, x, fa.FORWARD)
    run_plan_on_host(p, x, x)
    assert np.array_equal(x.reshape(n1, n0), x0.reshape(n0, n1).T)
    # 2-D in place with the two output axes exchanged
    a, b = 24, 40
    x = crand(rng2, 1, a * b).reshape(-1)
    x0 = x.copy()
    p = fa.plan_guru64_dft([(a, b, 1), (b, 1, a)], [], x, x, fa.BACKWARD)
    run_plan_on_host(p, x, x)
    want = oracle_dft(x0.reshape(1, -1), (a, b), 1, 1).reshape(a, b).T.reshape(-1)
    assert aerror(x, want) < TOL
    # different locations on the two sides stay rejected (A.c:4090-4094)
    with pytest.raises(ValueError):
        fa.plan_guru64_dft([(8, 1, 2)], [], x[:16], x[:16], fa.FORWARD)
