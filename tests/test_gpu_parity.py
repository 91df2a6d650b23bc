"""GPU parity tests: every transform goes through the C ABI of
libfftw3_amd.so onto device memory and is compared with the CPU oracle, the
golden fixtures and -- at the BASELINE sizes where no stored answer exists --
the reference verifier's properties (tests/verifier.py).  Tolerance: relative
L-infinity 1e-10, the reference's own bound for double precision
(fftw/libbench2/bench-main.c:70; north_star: "within 1e-10 rel-error")."""
import os

import numpy as np
import pytest

import fftw3_amd as fa
import verifier
from util import TOL, aerror, crand, oracle_c2r, oracle_dft, oracle_r2c, rrand

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert fa.device_count() > 0, "no HIP device: the GPU tier cannot run"
    return torch, torch.device("cuda:0")


def gpu_c2c(torch_dev, x, shape, b, sign=-1, inplace=False):
    torch, dev = torch_dev
    nn = int(np.prod(shape))
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    yd = xd if inplace else torch.zeros_like(xd)
    p = fa.plan_many_dft(len(shape), list(shape), b, xd, None, 1, nn, yd, None, 1, nn, sign)
    p.execute()
    torch.cuda.synchronize()
    return yd.cpu().numpy()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 13, 16, 17, 25, 31, 32, 61, 64, 77,
                               97, 100, 143, 1009, 1024, 1031, 4096, 5000, 15015, 17408, 65536,
                               65537, 1 << 20, 1 << 21])
def test_c2c_1d_matches_oracle(torch_dev, n):
    rng = np.random.default_rng(n)
    for b in (1, 3):
        for sign in (-1, 1):
            x = crand(rng, b, n)
            y = gpu_c2c(torch_dev, x, (n,), b, sign)
            assert aerror(y, oracle_dft(x, (n,), b, sign).reshape(b, n)) < TOL
    x = crand(rng, 2, n)
    y = gpu_c2c(torch_dev, x, (n,), 2, -1, inplace=True)
    assert aerror(y, oracle_dft(x, (n,), 2).reshape(2, n)) < TOL


@pytest.mark.parametrize("n", [1 << 22, 1920 * 2048, 2000 * 2000, 1080 * 1920, 2048 * 1280])
def test_two_trips_with_both_lengths_above_1024(torch_dev, n):
    """n = L1 x L2 with both lengths in 1025 ... 2048: two trips through the 512-item strided / transposed kernels
    (first pass down the columns, last pass rows in / transposed store with the twiddle on its input), where round 2
    took three; a batch that is not a multiple of anything, forward out of place and backward in place"""
    torch, dev = torch_dev
    rng = np.random.default_rng(n)
    b = 3
    x = crand(rng, b, n)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.zeros_like(xd)
    p = fa.plan_many_dft(1, [n], b, xd, None, 1, n, yd, None, 1, n, fa.FORWARD)
    lens = [s.L for s in p.steps()]
    assert len(lens) == 2 and min(lens) > 1024 and lens[0] * lens[1] == n, p.sprint()
    p.execute()
    torch.cuda.synchronize()
    assert aerror(yd.cpu().numpy(), oracle_dft(x, (n,), b).reshape(b, n)) < TOL
    y = gpu_c2c(torch_dev, x, (n,), b, +1, inplace=True)
    assert aerror(y, oracle_dft(x, (n,), b, +1).reshape(b, n)) < TOL


@pytest.mark.parametrize("n", [1 << 22, 1 << 21, 2048 * 1000, 2048 * 1920, 2048 * 256])
def test_r2c_in_two_trips(torch_dev, n, monkeypatch):
    """a long real transform n = L1 x 2048 decimated over the real data (emit_r2c_decimated): the complex pass of length
    L1 over the input read as pairs, then rows with the two real columns separated on the load side and the upper
    half stored conjugated at the mirrored index (FFTW_AMD_F_REAL_DEC) -- two trips where the half-length plans take
    three (a FFTW_MEASURE candidate, FFTW_AMD_REAL_DEC=1 here); out of place, in place (FFTW's padded layout), and
    against the three-trip plan"""
    torch, dev = torch_dev
    monkeypatch.setenv("FFTW_AMD_REAL_DEC", "1")     # off under FFTW_ESTIMATE: it only ties with the three-trip plans
    rng = np.random.default_rng(n)
    b = 3
    xr = rrand(rng, b, n)
    want = oracle_r2c(xr, (n,), b).reshape(b, n // 2 + 1)
    xd = torch.from_numpy(xr).to(dev)
    yd = torch.zeros(b, n // 2 + 1, dtype=torch.complex128, device=dev)
    p = fa.plan_many_dft_r2c(1, [n], b, xd, None, 1, n, yd, None, 1, n // 2 + 1)
    assert len(p.steps()) == 2 and "real-decimated" in p.sprint(), p.sprint()
    p.execute()
    torch.cuda.synchronize()
    got = yd.cpu().numpy()
    assert aerror(got, want) < TOL
    assert got[:, 0].imag.max() == 0.0 and got[:, n // 2].imag.max() == 0.0
    # in place: rows of n + 2 reals
    pd = torch.zeros(b, n + 2, dtype=torch.float64, device=dev)
    pd[:, :n] = xd
    q = fa.plan_many_dft_r2c(1, [n], b, pd, None, 1, n + 2, pd, None, 1, n // 2 + 1)
    assert "real-decimated" in q.sprint(), q.sprint()
    q.execute()
    torch.cuda.synchronize()
    assert aerror(torch.view_as_complex(pd.reshape(b, n // 2 + 1, 2)).cpu().numpy(), want) < TOL
    # the three-trip plan of the same transform agrees to rounding
    monkeypatch.delenv("FFTW_AMD_REAL_DEC")
    y3 = torch.zeros_like(yd)
    p3 = fa.plan_many_dft_r2c(1, [n], b, xd, None, 1, n, y3, None, 1, n // 2 + 1)
    assert "real-decimated" not in p3.sprint()
    p3.execute()
    torch.cuda.synchronize()
    assert aerror(y3.cpu().numpy(), want) < TOL


def test_sweep_1_to_100_and_pow2(torch_dev):
    """the reference's size sweep (fftw/tests/check.pl:126-173), forward and backward"""
    rng = np.random.default_rng(7)
    for n in list(range(1, 101)) + [128, 256, 512, 2048]:
        x = crand(rng, 2, n)
        for sign in (-1, 1):
            y = gpu_c2c(torch_dev, x, (n,), 2, sign)
            assert aerror(y, oracle_dft(x, (n,), 2, sign).reshape(2, n)) < TOL, n


def test_golden_fixtures(torch_dev):
    torch, dev = torch_dev
    z = np.load(os.path.join(GOLD, "c2c_1d.npz"))
    for k in [f for f in z.files if f.startswith("n") and f.endswith("_in")]:
        n = int(k[1:-3])
        x = z[k].reshape(1, n)
        assert aerror(gpu_c2c(torch_dev, x, (n,), 1, -1)[0], z["n%d_fwd" % n]) < TOL, n
        assert aerror(gpu_c2c(torch_dev, x, (n,), 1, +1)[0], z["n%d_bwd" % n]) < TOL, n
    for n in (4, 13, 64, 1024):
        x = z["v3n%d_in" % n]
        assert aerror(gpu_c2c(torch_dev, x, (n,), 3), z["v3n%d_fwd" % n]) < TOL
        # interleaved vectors NvV: stride 3, dist 1
        xi = torch.from_numpy(np.ascontiguousarray(x.T)).to(dev)
        yi = torch.zeros_like(xi)
        fa.plan_many_dft(1, [n], 3, xi, None, 3, 1, yi, None, 3, 1, fa.FORWARD).execute()
        torch.cuda.synchronize()
        assert aerror(yi.cpu().numpy().T, z["v3n%d_fwd" % n]) < TOL
    z = np.load(os.path.join(GOLD, "r2c_1d.npz"))
    for k in [f for f in z.files if f.endswith("_in")]:
        n = int(k[1:-3])
        xd = torch.from_numpy(z[k]).to(dev)
        yd = torch.zeros(n // 2 + 1, dtype=torch.complex128, device=dev)
        fa.plan_dft_r2c_1d(n, xd, yd).execute()
        torch.cuda.synchronize()
        assert aerror(yd.cpu().numpy(), z["n%d_out" % n]) < TOL, n
        zd = torch.zeros(n, dtype=torch.float64, device=dev)
        yin = torch.from_numpy(z["n%d_out" % n]).to(dev)
        fa.plan_dft_c2r_1d(n, yin, zd).execute()
        torch.cuda.synchronize()
        assert aerror(zd.cpu().numpy(), z[k] * n) < TOL, n
    z = np.load(os.path.join(GOLD, "nd.npz"))
    for k in [f for f in z.files if f.startswith("c") and f.endswith("_in")]:
        shape = tuple(int(s) for s in k[1:-3].split("x"))
        y = gpu_c2c(torch_dev, z[k].reshape((1,) + shape), shape, 1)
        assert aerror(y[0], z[k[:-3] + "_fwd"]) < TOL
    for k in [f for f in z.files if f.startswith("r") and f.endswith("_in")]:
        shape = tuple(int(s) for s in k[1:-3].split("x"))
        xd = torch.from_numpy(z[k]).to(dev)
        yd = torch.zeros((shape[0], shape[1] // 2 + 1), dtype=torch.complex128, device=dev)
        fa.plan_dft_r2c_2d(shape[0], shape[1], xd, yd).execute()
        torch.cuda.synchronize()
        assert aerror(yd.cpu().numpy(), z[k[:-3] + "_out"]) < TOL


def test_second_golden_set(torch_dev):
    """round 3 pins (tests/golden/pins2.npz, generator make_golden2.py: defining sums in 80-bit long double on
    the verifier's inputs), GPU against the fixtures directly: Rader 65537 / 12289, Bluestein 8191, 60060
    (sampled bins), odd-length r2c / c2r, 3-D c2c / r2c / c2r, every r2r kind at even and odd lengths"""
    torch, dev = torch_dev
    z = np.load(os.path.join(GOLD, "pins2.npz"))
    for n in (65537, 12289, 8191, 60060):
        x, bins = z["c%d_in" % n].reshape(1, n), z["c%d_bins" % n]
        scale = np.abs(z["c%d_fwd" % n]).max()
        assert np.abs(gpu_c2c(torch_dev, x, (n,), 1, -1)[0][bins] - z["c%d_fwd" % n]).max() < TOL * scale, n
        assert np.abs(gpu_c2c(torch_dev, x, (n,), 1, +1)[0][bins] - z["c%d_bwd" % n]).max() < TOL * scale, n
    for n in (77, 1001):
        xd = torch.from_numpy(z["r%d_in" % n]).to(dev)
        yd = torch.zeros(n // 2 + 1, dtype=torch.complex128, device=dev)
        fa.plan_dft_r2c_1d(n, xd, yd).execute()
        torch.cuda.synchronize()
        assert aerror(yd.cpu().numpy(), z["r%d_out" % n]) < TOL, n
        zd = torch.zeros(n, dtype=torch.float64, device=dev)
        yin = torch.from_numpy(z["r%d_out" % n]).to(dev)
        fa.plan_dft_c2r_1d(n, yin, zd).execute()
        torch.cuda.synchronize()
        assert aerror(zd.cpu().numpy(), z["r%d_in" % n] * n) < TOL, n
    for key in ("6x10x8", "5x6x7"):
        shape = tuple(int(v) for v in key.split("x"))
        y = gpu_c2c(torch_dev, z["c3_%s_in" % key].reshape((1,) + shape), shape, 1)
        assert aerror(y[0], z["c3_%s_fwd" % key]) < TOL
        xd = torch.from_numpy(z["r3_%s_in" % key]).to(dev)
        hs = shape[:2] + (shape[2] // 2 + 1,)
        yd = torch.zeros(hs, dtype=torch.complex128, device=dev)
        fa.plan_dft_r2c_3d(shape[0], shape[1], shape[2], xd, yd).execute()
        torch.cuda.synchronize()
        assert aerror(yd.cpu().numpy(), z["r3_%s_out" % key]) < TOL
        zd = torch.zeros(shape, dtype=torch.float64, device=dev)
        yin = torch.from_numpy(z["r3_%s_out" % key]).to(dev)
        fa.plan_dft_c2r_3d(shape[0], shape[1], shape[2], yin, zd).execute()
        torch.cuda.synchronize()
        assert aerror(zd.cpu().numpy(), z["r3_%s_in" % key] * np.prod(shape)) < TOL
    for n in (16, 15, 1000, 243):
        for kind in range(11):
            xd = torch.from_numpy(z["k%d_n%d_in" % (kind, n)]).to(dev)
            yd = torch.zeros(n, dtype=torch.float64, device=dev)
            fa.plan_r2r_1d(n, xd, yd, kind).execute()
            torch.cuda.synchronize()
            assert aerror(yd.cpu().numpy(), z["k%d_n%d_out" % (kind, n)]) < TOL, (kind, n)


@pytest.mark.parametrize("shape", [(4, 4), (8, 16), (16, 8), (13, 11), (64, 64), (3, 5, 7),
                                   (600, 6), (5, 2048), (256, 256), (30, 30), (2, 3, 4, 5),
                                   (1030, 4), (4096, 3)])
def test_c2c_nd_matches_oracle(torch_dev, shape):
    rng = np.random.default_rng(3)
    for b in (1, 2):
        for sign in (-1, 1):
            x = crand(rng, b, *shape)
            y = gpu_c2c(torch_dev, x, shape, b, sign)
            assert aerror(y, oracle_dft(x, shape, b, sign).reshape(x.shape)) < TOL


@pytest.mark.parametrize("n", [2, 3, 4, 8, 15, 16, 64, 128, 4096, 10000, 17, 97, 1009, 2018,
                               1 << 15, 1 << 21])
def test_r2c_c2r_1d(torch_dev, n):
    torch, dev = torch_dev
    rng = np.random.default_rng(n)
    b = 2
    x = rrand(rng, b, n)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.zeros((b, n // 2 + 1), dtype=torch.complex128, device=dev)
    fa.plan_many_dft_r2c(1, [n], b, xd, None, 1, n, yd, None, 1, n // 2 + 1).execute()
    torch.cuda.synchronize()
    ref = oracle_r2c(x, (n,), b).reshape(b, n // 2 + 1)
    y = yd.cpu().numpy()
    assert aerror(y, ref) < TOL
    assert np.all(y[:, 0].imag == 0) and (n % 2 or np.all(y[:, n // 2].imag == 0))
    assert np.array_equal(xd.cpu().numpy(), x)                 # r2c preserves its input
    yin = torch.from_numpy(ref).to(dev)
    zd = torch.zeros((b, n), dtype=torch.float64, device=dev)
    fa.plan_many_dft_c2r(1, [n], b, yin, None, 1, n // 2 + 1, zd, None, 1, n).execute()
    torch.cuda.synchronize()
    assert aerror(zd.cpu().numpy(), oracle_c2r(ref, (n,), b).reshape(b, n)) < TOL


@pytest.mark.parametrize("shape", [(4, 4), (8, 16), (13, 11), (16, 9), (64, 64), (5, 6, 8), (128, 1024)])
def test_r2c_c2r_nd(torch_dev, shape):
    torch, dev = torch_dev
    rng = np.random.default_rng(5)
    nn, hs = int(np.prod(shape)), shape[:-1] + (shape[-1] // 2 + 1,)
    hh = int(np.prod(hs))
    x = rrand(rng, 2, *shape)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.zeros((2,) + hs, dtype=torch.complex128, device=dev)
    fa.plan_many_dft_r2c(len(shape), list(shape), 2, xd, None, 1, nn, yd, None, 1, hh).execute()
    torch.cuda.synchronize()
    ref = oracle_r2c(x, shape, 2).reshape((2,) + hs)
    assert aerror(yd.cpu().numpy(), ref) < TOL
    yin = torch.from_numpy(ref).to(dev)
    zd = torch.zeros((2,) + shape, dtype=torch.float64, device=dev)
    fa.plan_many_dft_c2r(len(shape), list(shape), 2, yin, None, 1, hh, zd, None, 1, nn).execute()
    torch.cuda.synchronize()
    assert aerror(zd.cpu().numpy(), x * nn) < TOL
    assert np.array_equal(yin.cpu().numpy(), ref)              # scratch keeps the input intact


def test_radix4_real_transforms_forced(torch_dev, monkeypatch):
    """the rdft2-ct-dit/4 style plan (two quarter-length complex DFTs + radix-4 untangle)
    on sizes where the planner would not pick it by itself"""
    torch, dev = torch_dev
    monkeypatch.setenv("FFTW_AMD_FORCE_RADIX4", "1")
    rng = np.random.default_rng(44)
    for n in (8, 12, 20, 100, 1000, 4096, 40000, 1 << 16):
        b = 3
        x = rrand(rng, b, n)
        xd = torch.from_numpy(x).to(dev)
        yd = torch.zeros((b, n // 2 + 1), dtype=torch.complex128, device=dev)
        p = fa.plan_many_dft_r2c(1, [n], b, xd, None, 1, n, yd, None, 1, n // 2 + 1)
        assert "untangle4" in p.sprint()
        p.execute()
        torch.cuda.synchronize()
        ref = oracle_r2c(x, (n,), b).reshape(b, n // 2 + 1)
        assert aerror(yd.cpu().numpy(), ref) < TOL, n
        yin = torch.from_numpy(ref).to(dev)
        zd = torch.zeros((b, n), dtype=torch.float64, device=dev)
        p = fa.plan_many_dft_c2r(1, [n], b, yin, None, 1, n // 2 + 1, zd, None, 1, n)
        assert "tangle4" in p.sprint()
        p.execute()
        torch.cuda.synchronize()
        assert aerror(zd.cpu().numpy(), x * n) < TOL, n


@pytest.mark.parametrize("n", [64, 128, 256, 512, 1024, 2048, 4096, 8192, 1 << 14, 1 << 18, 1 << 21])
def test_register_kernels_wide_batches(torch_dev, n):
    """every register-kernel length with batches wide enough to fill their tiles,
    including ragged tile tails, forward / backward / in place"""
    rng = np.random.default_rng(n + 1)
    for b in ((200, 77) if n <= 4096 else (5,)):
        for sign in (-1, 1):
            x = crand(rng, b, n)
            y = gpu_c2c(torch_dev, x, (n,), b, sign)
            assert aerror(y, oracle_dft(x, (n,), b, sign).reshape(b, n)) < TOL
    x = crand(rng, 9, n)
    y = gpu_c2c(torch_dev, x, (n,), 9, -1, inplace=True)
    assert aerror(y, oracle_dft(x, (n,), 9).reshape(9, n)) < TOL


@pytest.mark.parametrize("n", [8192, 16384])
def test_rows_of_8192_and_16384_in_one_trip(torch_dev, n):
    """contiguous interleaved rows of 8192 / 16384 points: one workgroup per row (32 x 16 x 16 with 256 items,
    pass3s.hpp; 32 x 16 x 32 with 512 items, pass3w.hpp), steps without an LDS-kernel fallback -- so plans that
    cannot promise the layout at plan time (FFTW_UNALIGNED, a lone transform) keep the two-pass split; both
    compute the same answer"""
    torch, dev = torch_dev
    rng = np.random.default_rng(n)
    for b, sign in ((77, -1), (3, 1)):
        x = crand(rng, b, n)
        dx = torch.from_numpy(x).to(dev)
        dy = torch.zeros_like(dx)
        p = fa.plan_many_dft(1, [n], b, dx, None, 1, n, dy, None, 1, n, sign)
        assert "pass-%d/reg3" % n in p.sprint() and len(p.steps()) == 1, p.sprint()
        p.execute()
        p.sync()
        ref = oracle_dft(x, (n,), b, sign).reshape(b, n)
        assert aerror(dy.cpu().numpy(), ref) < TOL
        q = fa.plan_many_dft(1, [n], b, dx, None, 1, n, dy, None, 1, n, sign, fa.ESTIMATE | fa.UNALIGNED)
        assert len(q.steps()) == 2, q.sprint()
        dy.zero_()
        q.execute()
        q.sync()
        assert aerror(dy.cpu().numpy(), ref) < TOL
        # in place: every row maps onto itself
        r = fa.plan_many_dft(1, [n], b, dx, None, 1, n, dx, None, 1, n, sign)
        assert len(r.steps()) == 1, r.sprint()
        r.execute()
        r.sync()
        assert aerror(dx.cpu().numpy(), ref) < TOL
    # padded rows (idist > n) and the rows axis of a 2-D transform
    x = crand(rng, 5, n + 6)
    dx = torch.from_numpy(x).to(dev)
    dy = torch.zeros(5, n + 2, dtype=torch.complex128, device=dev)
    p = fa.plan_many_dft(1, [n], 5, dx, None, 1, n + 6, dy, None, 1, n + 2, fa.FORWARD)
    assert "pass-%d/reg3" % n in p.sprint(), p.sprint()
    p.execute()
    p.sync()
    assert aerror(dy.cpu().numpy()[:, :n], oracle_dft(np.ascontiguousarray(x[:, :n]), (n,), 5).reshape(5, n)) < TOL
    assert float(dy[:, n:].abs().max()) == 0.0
    x = crand(rng, 1, 12 * n)
    y = gpu_c2c(torch_dev, x, (12, n), 1)
    assert aerror(y, oracle_dft(x, (12, n), 1).reshape(1, 12 * n)) < TOL
    one = fa.plan_dft_1d(n, dx, dy, fa.FORWARD)
    assert len(one.steps()) == 2, one.sprint()


@pytest.mark.parametrize("k,lens", [(22, [2048, 2048]), (23, [128, 128, 512]), (24, [128, 128, 1024]),
                                    (25, [512, 128, 512])])
def test_three_pass_powers_of_two(torch_dev, k, lens):
    """2^23 ... 2^25: the three-pass split comes from the planner's per-position cost table
    (pow2_three_pass_split); 2^22 = 2048 x 2048 takes two trips on the 512-item kernels since round 3; forward c2c
    against the oracle, and r2c of twice the length"""
    torch, dev = torch_dev
    n = 1 << k
    rng = np.random.default_rng(k)
    b = 2 if k <= 23 else 1
    x = crand(rng, b, n)
    dx = torch.from_numpy(x).to(dev)
    dy = torch.zeros_like(dx)
    p = fa.plan_many_dft(1, [n], b, dx, None, 1, n, dy, None, 1, n, fa.FORWARD)
    assert [s.L for s in p.steps()] == lens, p.sprint()
    p.execute()
    p.sync()
    assert aerror(dy.cpu().numpy(), oracle_dft(x, (n,), b).reshape(b, n)) < TOL
    if k <= 23:
        xr = rrand(rng, 2 * n)
        dr = torch.from_numpy(xr).to(dev)
        dc = torch.zeros(n + 1, dtype=torch.complex128, device=dev)
        q = fa.plan_dft_r2c_1d(2 * n, dr, dc)
        q.execute()
        q.sync()
        assert aerror(dc.cpu().numpy(), oracle_r2c(xr, (2 * n,))) < TOL


@pytest.mark.parametrize("k", [26, 28])
def test_huge_single_transform_known_answer(torch_dev, k):
    """one transform of 2^26 / 2^28 points (1 / 4 GiB): too long for the oracle, so the check is the
    shifted-impulse known answer X[k] = exp(-2 pi i j0 k / n), evaluated on the device with the
    index product reduced exactly in int64 (the reference's impulse test, verify-lib.c:285-340)"""
    import math
    torch, dev = torch_dev
    n = 1 << k
    j0 = 123456789 % n
    x = torch.zeros(n, dtype=torch.complex128, device=dev)
    x[j0] = 1.0
    y = torch.zeros_like(x)
    p = fa.plan_dft_1d(n, x, y, fa.FORWARD)
    assert len(p.steps()) == 3, p.sprint()
    p.execute()
    p.sync()
    worst = 0.0
    step = 1 << 24
    for s in range(0, n, step):
        kk = torch.arange(s, min(n, s + step), dtype=torch.int64, device=dev)
        ang = ((kk * j0) % n).to(torch.float64) * (-2.0 * math.pi / n)
        worst = max(worst, (y[s:s + step] - torch.complex(torch.cos(ang), torch.sin(ang))).abs().max().item())
    assert worst < 1e-13, worst
    del p, x, y
    torch.cuda.empty_cache()


def test_layouts_inplace_embed_split_guru_newarray(torch_dev):
    torch, dev = torch_dev
    rng = np.random.default_rng(9)
    # in-place padded r2c (fftw_rdft2_pad layout)
    n, b = 1000, 3
    pad = 2 * (n // 2 + 1)
    x = rrand(rng, b, n)
    buf = np.zeros((b, pad))
    buf[:, :n] = x
    bd = torch.from_numpy(buf).to(dev)
    fa.plan_many_dft_r2c(1, [n], b, bd, None, 1, pad, bd, None, 1, pad // 2).execute()
    torch.cuda.synchronize()
    assert aerror(bd.cpu().numpy().view(complex), oracle_r2c(x, (n,), b).reshape(b, n // 2 + 1)) < TOL
    # embedded sub-array, untouched surroundings
    xe = crand(rng, 12, 20)
    xd = torch.from_numpy(xe).to(dev)
    yd = torch.full((10, 16), 7.0 + 0j, dtype=torch.complex128, device=dev)
    fa.plan_many_dft(2, [6, 8], 1, xd, [12, 20], 1, 0, yd, [10, 16], 1, 0, fa.FORWARD).execute()
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    assert aerror(y[:6, :8], np.fft.fft2(xe[:6, :8])) < TOL
    assert np.all(y[6:] == 7) and np.all(y[:, 8:] == 7)
    # guru: strided transform dim, two loop dims
    n, h0, h1 = 30, 3, 2
    xg = crand(rng, h0, h1, n * 7)
    xd = torch.from_numpy(xg).to(dev)
    yd = torch.zeros((h0, h1, n), dtype=torch.complex128, device=dev)
    fa.plan_guru64_dft([(n, 7, 1)], [(h0, h1 * n * 7, h1 * n), (h1, n * 7, n)], xd, yd,
                       fa.FORWARD).execute()
    torch.cuda.synchronize()
    assert aerror(yd.cpu().numpy(), np.fft.fft(xg[:, :, ::7], axis=2)) < TOL
    # split arrays
    n = 960
    planes = torch.from_numpy(rrand(rng, 2, n)).to(dev)
    outp = torch.zeros_like(planes)
    p = fa.plan_guru64_split_dft([(n, 1, 1)], [], planes[0], planes[1], outp[0], outp[1])
    p.execute()
    torch.cuda.synchronize()
    pl = planes.cpu().numpy()
    o = outp.cpu().numpy()
    assert aerror(o[0] + 1j * o[1], np.fft.fft(pl[0] + 1j * pl[1])) < TOL
    # new-array execute on other buffers (reference fftw_execute_dft, A.c:434-440)
    n, b = 4096, 4
    x1, x2 = crand(rng, b, n), crand(rng, b, n)
    d1, d2 = torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev)
    o1, o2 = torch.zeros_like(d1), torch.zeros_like(d2)
    p = fa.plan_many_dft(1, [n], b, d1, None, 1, n, o1, None, 1, n, fa.FORWARD)
    p.execute_dft(d2, o2)
    torch.cuda.synchronize()
    assert aerror(o2.cpu().numpy(), oracle_dft(x2, (n,), b).reshape(b, n)) < TOL
    assert float(o1.abs().max()) == 0.0
    # howmany == 0 executes nothing
    fa.plan_many_dft(1, [n], 0, d1, None, 1, n, o1, None, 1, n, fa.FORWARD).execute()


def test_measure_mode_records_wisdom(torch_dev):
    """FFTW_MEASURE times candidate configurations on the device (it may overwrite the
    arrays, like the reference), the winner becomes wisdom, and a later
    FFTW_WISDOM_ONLY plan of the same problem reuses it and computes the same answer"""
    torch, dev = torch_dev
    fa.forget_wisdom()
    n, b = 1 << 16, 64
    rng = np.random.default_rng(8)
    x = crand(rng, b, n)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.zeros_like(xd)
    p = fa.plan_many_dft(1, [n], b, xd, None, 1, n, yd, None, 1, n, fa.FORWARD, fa.MEASURE)
    text = fa.export_wisdom_to_string()
    assert "65536:2:2" in text
    xd.copy_(torch.from_numpy(x))
    p.execute()
    torch.cuda.synchronize()
    ref = oracle_dft(x, (n,), b).reshape(b, n)
    assert aerror(yd.cpu().numpy(), ref) < TOL
    fa.forget_wisdom()
    with pytest.raises(ValueError):
        fa.plan_many_dft(1, [n], b, xd, None, 1, n, yd, None, 1, n, fa.FORWARD, fa.ESTIMATE | fa.WISDOM_ONLY)
    assert fa.import_wisdom_from_string(text) == 1
    p2 = fa.plan_many_dft(1, [n], b, xd, None, 1, n, yd, None, 1, n, fa.FORWARD, fa.ESTIMATE | fa.WISDOM_ONLY)
    yd.zero_()
    p2.execute()
    torch.cuda.synchronize()
    assert aerror(yd.cpu().numpy(), ref) < TOL
    fa.forget_wisdom()


def test_measure_mode_times_the_decimated_r2c_plan_too(torch_dev):
    """for r2c problems FFTW_MEASURE also times the plan decimated over the real data (cfg.real_dec); whichever wins,
    the wisdom round-trips (the candidate is bit 16 of the wisdom line's flag field) and both plans compute the
    oracle's answer"""
    torch, dev = torch_dev
    fa.forget_wisdom()
    n, b = 1 << 20, 24
    rng = np.random.default_rng(21)
    xr = rrand(rng, b, n)
    want = oracle_r2c(xr, (n,), b).reshape(b, n // 2 + 1)
    xd = torch.from_numpy(xr).to(dev)
    yd = torch.zeros(b, n // 2 + 1, dtype=torch.complex128, device=dev)
    p = fa.plan_many_dft_r2c(1, [n], b, xd, None, 1, n, yd, None, 1, n // 2 + 1, fa.MEASURE)
    text = fa.export_wisdom_to_string()
    picked = "real-decimated" in p.sprint()
    xd.copy_(torch.from_numpy(xr))
    p.execute()
    torch.cuda.synchronize()
    assert aerror(yd.cpu().numpy(), want) < TOL
    fa.forget_wisdom()
    assert fa.import_wisdom_from_string(text) == 1
    p2 = fa.plan_many_dft_r2c(1, [n], b, xd, None, 1, n, yd, None, 1, n // 2 + 1, fa.ESTIMATE | fa.WISDOM_ONLY)
    assert ("real-decimated" in p2.sprint()) == picked, (p.sprint(), p2.sprint(), text)
    yd.zero_()
    p2.execute()
    torch.cuda.synchronize()
    assert aerror(yd.cpu().numpy(), want) < TOL
    fa.forget_wisdom()


def test_host_arrays_are_staged(torch_dev):
    """plain host pointers (numpy) take the PCIe staging path, incl. gaps in the output"""
    rng = np.random.default_rng(2)
    x = crand(rng, 2, 1000)
    y = np.zeros_like(x)
    fa.plan_many_dft(1, [1000], 2, x, None, 1, 1000, y, None, 1, 1000, fa.FORWARD).execute()
    assert aerror(y, oracle_dft(x, (1000,), 2).reshape(2, 1000)) < TOL
    xs = crand(rng, 64 * 3)
    ys = np.full(64 * 2, 5.0 + 0j)
    fa.plan_many_dft(1, [64], 1, xs, None, 3, 0, ys, None, 2, 0, fa.BACKWARD).execute()
    assert aerror(ys[::2], np.fft.ifft(xs[::3]) * 64) < TOL
    assert np.all(ys[1::2] == 5.0)
    xi = x.copy()
    fa.plan_many_dft(1, [1000], 2, xi, None, 1, 1000, xi, None, 1, 1000, fa.FORWARD).execute()
    assert aerror(xi, y) < TOL
    xr = rrand(rng, 500)
    yr = np.zeros(251, dtype=complex)
    fa.plan_dft_r2c_1d(500, xr, yr).execute()
    assert aerror(yr, np.fft.rfft(xr)) < TOL


def test_chunked_batches_and_ragged_tail(torch_dev):
    torch, dev = torch_dev
    rng = np.random.default_rng(4)
    fa.set_chunk_bytes(1 << 20)
    try:
        n, b = 1 << 15, 11                     # 512 KiB per transform -> chunk 2, tail 1 (2^14 runs in one trip: no scratch)
        x = crand(rng, b, n)
        xd = torch.from_numpy(x).to(dev)
        yd = torch.zeros_like(xd)
        p = fa.plan_many_dft(1, [n], b, xd, None, 1, n, yd, None, 1, n, fa.FORWARD)
        assert 1 <= p.chunk < b
        p.execute()
        torch.cuda.synchronize()
        assert aerror(yd.cpu().numpy(), oracle_dft(x, (n,), b).reshape(b, n)) < TOL
    finally:
        fa.set_chunk_bytes(0)


@pytest.mark.parametrize("pipeline", ["0", "1"])
def test_chunk_pipeline_on_and_off(torch_dev, monkeypatch, pipeline):
    """the two-stream chunk pipeline (a FFTW_MEASURE candidate, FFTW_AMD_PIPELINE=1) and the
    serial default give the same transforms: c2c two-pass, r2c, r2r, in place and out of place"""
    from util import oracle_r2c, oracle_r2r, rrand
    torch, dev = torch_dev
    rng = np.random.default_rng(40)
    monkeypatch.setenv("FFTW_AMD_PIPELINE", pipeline)
    fa.set_chunk_bytes(8 << 20)
    try:
        n, b = 1 << 16, 37                     # 1 MiB per transform: chunk 4 .. 8, ragged tail
        x = crand(rng, b, n)
        xd = torch.from_numpy(x).to(dev)
        yd = torch.zeros_like(xd)
        p = fa.plan_many_dft(1, [n], b, xd, None, 1, n, yd, None, 1, n, fa.FORWARD)
        assert p.chunk < b
        p.execute()
        torch.cuda.synchronize()
        assert aerror(yd.cpu().numpy(), oracle_dft(x, (n,), b).reshape(b, n)) < TOL
        q = fa.plan_many_dft(1, [n], b, xd, None, 1, n, xd, None, 1, n, fa.FORWARD)   # in place
        q.execute()
        torch.cuda.synchronize()
        assert aerror(xd.cpu().numpy(), oracle_dft(x, (n,), b).reshape(b, n)) < TOL
        xr = rrand(rng, b, n)
        xrd = torch.from_numpy(xr).to(dev)
        yrd = torch.zeros((b, n // 2 + 1), dtype=torch.complex128, device=dev)
        r = fa.plan_many_dft_r2c(1, [n], b, xrd, None, 1, n, yrd, None, 1, n // 2 + 1)
        r.execute()
        torch.cuda.synchronize()
        assert aerror(yrd.cpu().numpy(), oracle_r2c(xr, (n,), b).reshape(b, n // 2 + 1)) < TOL
        zrd = torch.zeros_like(xrd)
        t = fa.plan_many_r2r(1, [n], b, xrd, None, 1, n, zrd, None, 1, n, [fa.REDFT10])
        t.execute()
        torch.cuda.synchronize()
        assert aerror(zrd.cpu().numpy().reshape(-1), oracle_r2r(xr.reshape(-1), [n], [fa.REDFT10], howmany=b)) < TOL
    finally:
        fa.set_chunk_bytes(0)
        monkeypatch.setenv("FFTW_AMD_PIPELINE", "0")
        z = np.zeros(4, dtype=np.complex128)
        fa.plan_dft_1d(4, z, z, fa.FORWARD)        # re-reads the environment


def _gpu_apply(torch_dev, shape, sign=-1):
    torch, dev = torch_dev
    cache = {}

    def apply(x):
        b = x.shape[0]
        xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        yd = torch.empty_like(xd)
        nn = int(np.prod(shape))
        if b not in cache:
            cache[b] = fa.plan_many_dft(len(shape), list(shape), b, xd, None, 1, nn, yd, None, 1,
                                        nn, sign)
        cache[b].execute_dft(xd, yd)
        torch.cuda.synchronize()
        return yd.cpu().numpy()
    return apply


@pytest.mark.parametrize("shape,vecn", [((1024,), 4), ((1 << 20,), 2), ((15015,), 2),
                                        ((1031,), 2), ((65537,), 1), ((256, 256), 2),
                                        ((4096, 4096), 1)])
def test_reference_verifier_on_gpu(torch_dev, shape, vecn):
    """impulse / linearity / shift battery of the reference (verify-lib.c) on the GPU path"""
    rounds = 1 if int(np.prod(shape)) > (1 << 22) else 3
    for sign in (-1, 1):
        e = verifier.verify_c2c(_gpu_apply(torch_dev, shape, sign), shape, vecn=vecn, sign=sign,
                                rounds=rounds)
        assert e < TOL


def test_square_4096_in_two_trips(torch_dev, monkeypatch):
    """cfg5 unit (round 3): 4096 x 4096 as TWO trips -- the strided 1024-point pass over the four row classes,
    then four rows per workgroup with the radix-4 butterfly across them (pass3q.hpp, FFTW_AMD_F_LO_DFT) --
    against the oracle and against the three-trip plan (FFTW_AMD_NO_LO_DFT): both signs, a batch that runs in
    several chunks on two lanes, in place, padded row pitch on both sides, new-array execute."""
    torch, dev = torch_dev
    n = 4096
    rng = np.random.default_rng(4096)
    x = crand(rng, 3, n * n)
    xd = torch.from_numpy(x).to(dev)
    want = {}
    for sign in (-1, 1):
        yd = torch.zeros_like(xd)
        p = fa.plan_many_dft(2, [n, n], 3, xd, None, 1, n * n, yd, None, 1, n * n, sign)
        assert len(p.steps()) == 2 and "dft4-across-rows" in p.sprint(), p.sprint()
        p.execute()
        torch.cuda.synchronize()
        got = yd.cpu().numpy()
        want[sign] = oracle_dft(x[:1], (n, n), 1, sign).reshape(1, n * n)
        assert aerror(got[:1], want[sign]) < TOL
        # the other images against the three-trip plan of round 2
        monkeypatch.setenv("FFTW_AMD_NO_LO_DFT", "1")
        zd = torch.zeros_like(xd)
        q = fa.plan_many_dft(2, [n, n], 3, xd, None, 1, n * n, zd, None, 1, n * n, sign)
        assert len(q.steps()) == 3, q.sprint()
        q.execute()
        torch.cuda.synchronize()
        monkeypatch.delenv("FFTW_AMD_NO_LO_DFT")
        assert aerror(got, zd.cpu().numpy()) < TOL
        del zd, yd
    # in place + new-array execute on another buffer
    wd = xd[:1].clone()
    r = fa.plan_many_dft(2, [n, n], 1, wd, None, 1, n * n, wd, None, 1, n * n, fa.FORWARD)
    assert len(r.steps()) == 2
    r.execute()
    torch.cuda.synchronize()
    assert aerror(wd.cpu().numpy(), want[-1]) < TOL
    w2 = xd[:1].clone()
    r.execute_dft(w2, w2)
    torch.cuda.synchronize()
    assert aerror(w2.cpu().numpy(), want[-1]) < TOL
    del wd, w2
    # rows padded to 4100 on the input side and 4104 on the output side (inembed / onembed)
    xp = torch.zeros(n, n + 4, dtype=torch.complex128, device=dev)
    xp[:, :n] = xd[0].view(n, n)
    yp = torch.zeros(n, n + 8, dtype=torch.complex128, device=dev)
    e = fa.plan_many_dft(2, [n, n], 1, xp, [n, n + 4], 1, n * (n + 4), yp, [n, n + 8], 1, n * (n + 8), fa.FORWARD)
    assert len(e.steps()) == 2, e.sprint()
    e.execute()
    torch.cuda.synchronize()
    assert aerror(yp[:, :n].contiguous().cpu().numpy().reshape(1, n * n), want[-1]) < TOL
    assert float(yp[:, n:].abs().max()) == 0.0


@pytest.mark.parametrize("n0,n1,t", [(2048, 2048, 4), (2048, 4096, 2), (4096, 2048, 4), (1536, 2048, 4), (1280, 4096, 2),
                                     (3072, 4096, 4)])
def test_rows_with_a_dft_across_the_rows_of_a_tile(torch_dev, n0, n1, t):
    """two-dimensional transforms whose strided axis is n0 = T x L0 (1024 < n0 <= 4096) over rows of 2048 / 4096
    points: strided L0-point pass with the twiddle w_n0^(s k'), then tiles of T rows with the DFT-T across them
    (pass3s_kernel XROW forms, pass3q.hpp; planner emit_rows_lo_dft).  Both signs, batch of 3, against the oracle."""
    torch, dev = torch_dev
    rng = np.random.default_rng(n0 + n1)
    x = crand(rng, 3, n0 * n1)
    xd = torch.from_numpy(x).to(dev)
    for sign in (-1, 1):
        yd = torch.zeros_like(xd)
        p = fa.plan_many_dft(2, [n0, n1], 3, xd, None, 1, n0 * n1, yd, None, 1, n0 * n1, sign)
        assert len(p.steps()) == 2 and ("dft%d-across-rows" % t) in p.sprint(), p.sprint()
        p.execute()
        torch.cuda.synchronize()
        assert aerror(yd.cpu().numpy(), oracle_dft(x, (n0, n1), 3, sign).reshape(3, n0 * n1)) < TOL


def test_inplace_with_different_strides_on_the_two_sides(torch_dev):
    """in-place problems whose input and output strides differ but address the same locations (the reference's
    fftw_tensor_inplace_locations rule, A.c:17298-17311; its own plans: dft-ct-dif + "q1_r" square codelets,
    A.c:2204-2250): transposed-output batched transforms, a rank-0 transpose and a 2-D transform with exchanged
    output axes, on the GPU against the oracle; also on plain host arrays (staged)."""
    torch, dev = torch_dev
    rng = np.random.default_rng(171)
    for n, v in ((64, 64), (48, 80), (1024, 16), (4096, 3), (1 << 16, 4)):
        x = crand(rng, 1, n * v).reshape(-1)
        xd = torch.from_numpy(x).to(dev)
        p = fa.plan_guru64_dft([(n, 1, v)], [(v, n, 1)], xd, xd, fa.FORWARD)
        p.execute()
        torch.cuda.synchronize()
        want = oracle_dft(x.reshape(1, -1), (n,), v).reshape(v, n).T.reshape(-1)
        assert aerror(xd.cpu().numpy(), want) < TOL, (n, v)
        # new-array execution on another in-place buffer
        x2 = crand(rng, 1, n * v).reshape(-1)
        x2d = torch.from_numpy(x2).to(dev)
        p.execute_dft(x2d, x2d)
        torch.cuda.synchronize()
        assert aerror(x2d.cpu().numpy(), oracle_dft(x2.reshape(1, -1), (n,), v).reshape(v, n).T.reshape(-1)) < TOL
    n0, n1 = 1200, 2000
    x = crand(rng, 1, n0 * n1).reshape(-1)
    xd = torch.from_numpy(x).to(dev)
    fa.plan_guru64_dft([], [(n0, n1, 1), (n1, 1, n0)], xd, xd, fa.FORWARD).execute()
    torch.cuda.synchronize()
    assert np.array_equal(xd.cpu().numpy().reshape(n1, n0), x.reshape(n0, n1).T)
    a, b = 240, 400
    x = crand(rng, 1, a * b).reshape(-1)
    xh = x.copy()                                                    # host array: staged in place
    fa.plan_guru64_dft([(a, b, 1), (b, 1, a)], [], xh, xh, fa.BACKWARD).execute()
    want = oracle_dft(x.reshape(1, -1), (a, b), 1, 1).reshape(a, b).T.reshape(-1)
    assert aerror(xh, want) < TOL


def test_mixed_radix_baseline_size_against_oracle(torch_dev):
    """cfg4 unit: N = 3*5*7*11*13*2^10, one transform against the oracle"""
    n = 3 * 5 * 7 * 11 * 13 * 1024
    rng = np.random.default_rng(15)
    x = crand(rng, 1, n)
    y = gpu_c2c(torch_dev, x, (n,), 1)
    assert aerror(y[0], oracle_dft(x, (n,))) < TOL


def test_r2c_baseline_size_roundtrip(torch_dev):
    """cfg3 unit: r2c N = 2^22 against the oracle, then c2r(r2c(x)) = N x"""
    torch, dev = torch_dev
    n, b = 1 << 22, 2
    rng = np.random.default_rng(22)
    x = rrand(rng, b, n)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.zeros((b, n // 2 + 1), dtype=torch.complex128, device=dev)
    fa.plan_many_dft_r2c(1, [n], b, xd, None, 1, n, yd, None, 1, n // 2 + 1).execute()
    torch.cuda.synchronize()
    assert aerror(yd.cpu().numpy(), oracle_r2c(x, (n,), b).reshape(b, n // 2 + 1)) < TOL
    zd = torch.zeros_like(xd)
    fa.plan_many_dft_c2r(1, [n], b, yd, None, 1, n // 2 + 1, zd, None, 1, n).execute()
    torch.cuda.synchronize()
    assert aerror(zd.cpu().numpy(), x * n) < TOL


def test_full_batch_known_answer_cfg2(torch_dev):
    """BASELINE configs[1] at full size: N = 2^20, howmany = 4096 (64 GiB in, 64 GiB
    out).  Row b holds an impulse (1+2i) at position s_b; its transform is the
    closed form (1+2i) exp(-2 pi i s_b k / N), checked on device for every row."""
    torch, dev = torch_dev
    n, b = 1 << 20, 4096
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip("needs ~140 GiB of free HBM")
    xd = torch.zeros((b, n), dtype=torch.complex128, device=dev)
    yd = torch.empty_like(xd)
    s = (torch.arange(b, device=dev, dtype=torch.int64) * 977 + 5) % n
    xd[torch.arange(b, device=dev), s] = 1 + 2j
    p = fa.plan_many_dft(1, [n], b, xd, None, 1, n, yd, None, 1, n, fa.FORWARD)
    p.execute()
    torch.cuda.synchronize()
    k = torch.arange(n, device=dev, dtype=torch.int64)
    worst = 0.0
    for r0 in range(0, b, 64):
        rows = slice(r0, r0 + 64)
        m = (s[rows, None] * k[None, :]) % n
        ang = m.to(torch.float64) * (-2.0 * np.pi / n)
        want = torch.polar(torch.ones_like(ang), ang) * (1 + 2j)
        worst = max(worst, float((yd[rows] - want).abs().max()))
        del m, ang, want
    assert worst / 2.0 < TOL
    del xd, yd
    torch.cuda.empty_cache()


def test_execute_is_capturable_in_a_hip_graph(torch_dev):
    """fftw_execute allocates nothing and forks/joins its side streams with events, so a
    caller can capture it into a HIP graph and replay it (new inputs in the same buffers)"""
    torch, dev = torch_dev
    n, b = 1 << 16, 40                    # two passes, several chunks -> the two-stream pipeline is active
    fa.set_chunk_bytes(8 << 20)
    try:
        rng = np.random.default_rng(77)
        x1, x2 = crand(rng, b, n), crand(rng, b, n)
        xd = torch.from_numpy(x1).to(dev)
        yd = torch.zeros_like(xd)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            p = fa.plan_many_dft(1, [n], b, xd, None, 1, n, yd, None, 1, n, fa.FORWARD)
            p.set_stream(s.cuda_stream)
            p.execute()                   # warm: everything allocated before the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            p.execute()
        xd.copy_(torch.from_numpy(x2))
        yd.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert aerror(yd.cpu().numpy(), oracle_dft(x2, (n,), b).reshape(b, n)) < TOL
        xd.copy_(torch.from_numpy(x1))
        g.replay()
        torch.cuda.synchronize()
        assert aerror(yd.cpu().numpy(), oracle_dft(x1, (n,), b).reshape(b, n)) < TOL
    finally:
        fa.set_chunk_bytes(0)


def test_random_composite_shapes_like_check_pl(torch_dev):
    """the reference's random sweep (fftw/tests/check.pl:186-251): random ranks up to 4,
    dimensions built from factors <= 13, random vector type (none / contiguous *V /
    interleaved vV), forward or backward, in or out of place; every case against the oracle"""
    torch, dev = torch_dev
    rng = np.random.default_rng(1234)
    primes = [2, 3, 5, 7, 11, 13]
    for case in range(60):
        rank = int(rng.integers(1, 5))
        shape = []
        budget = 200000 ** (1.0 / rank)
        for _ in range(rank):
            n = 1
            while True:
                f = primes[int(rng.integers(0, len(primes)))]
                if n * f > max(2.0, budget):
                    break
                n *= f
                if rng.random() < 0.25:
                    break
            shape.append(n)
        shape = tuple(shape)
        nn = int(np.prod(shape))
        vtype = int(rng.integers(0, 3))
        v = 1 if vtype == 0 else int(rng.integers(2, 5))
        sign = -1 if rng.random() < 0.5 else 1
        inplace = rng.random() < 0.5
        if vtype == 2:                               # interleaved vectors: stride v, dist 1
            x = crand(rng, *(shape + (v,)))
            istride, idist = v, 1
        else:
            x = crand(rng, *((v,) + shape))
            istride, idist = 1, nn
        xd = torch.from_numpy(x).to(dev)
        yd = xd if inplace else torch.zeros_like(xd)
        p = fa.plan_many_dft(rank, list(shape), v, xd, None, istride, idist, yd, None, istride, idist, sign)
        p.execute()
        torch.cuda.synchronize()
        ref = oracle_dft(x, shape, v, sign, istride=istride, idist=idist, ostride=istride, odist=idist)
        e = aerror(yd.cpu().numpy().reshape(-1), ref)
        assert e < TOL, (case, shape, v, vtype, sign, inplace, e, p.sprint())


C_CLIENT_RUN = r"""
/* an unmodified FFTW program: host arrays from fftw_malloc, plan, execute, check */
#include <math.h>
#include <stdio.h>
#include <fftw3.h>
int main(void) {
    int n = 4096, k, bad = 0;
    fftw_complex *in = fftw_alloc_complex(n), *out = fftw_alloc_complex(n);
    double *r = fftw_alloc_real(n), *d = fftw_alloc_real(n);
    fftw_plan p = fftw_plan_dft_1d(n, in, out, FFTW_FORWARD, FFTW_ESTIMATE);
    fftw_plan q = fftw_plan_dft_1d(n, out, in, FFTW_BACKWARD, FFTW_ESTIMATE);
    fftw_plan t = fftw_plan_r2r_1d(n, r, d, FFTW_REDFT10, FFTW_ESTIMATE);
    if (!p || !q || !t) return 2;
    for (k = 0; k < n; ++k) { in[k][0] = cos(2 * M_PI * 5 * k / n); in[k][1] = sin(2 * M_PI * 5 * k / n); r[k] = 1.0; }
    fftw_execute(p);                               /* a single tone lands in bin 5 */
    for (k = 0; k < n; ++k) {
        double wr = (k == 5) ? n : 0.0;
        if (fabs(out[k][0] - wr) > 1e-9 * n || fabs(out[k][1]) > 1e-9 * n) ++bad;
    }
    fftw_execute(q);                               /* and back: n times the input */
    for (k = 0; k < n; ++k)
        if (fabs(in[k][0] - n * cos(2 * M_PI * 5 * k / n)) > 1e-9 * n) ++bad;
    fftw_execute(t);                               /* DCT-II of a constant: 2n at k = 0 */
    for (k = 0; k < n; ++k)
        if (fabs(d[k] - (k == 0 ? 2.0 * n : 0.0)) > 1e-9 * n) ++bad;
    fftw_destroy_plan(p); fftw_destroy_plan(q); fftw_destroy_plan(t);
    fftw_free(in); fftw_free(out); fftw_free(r); fftw_free(d);
    printf(bad ? "client FAILED %d\n" : "client ok %d\n", bad);
    return bad != 0;
}
"""


def test_stock_c_client_runs_on_the_gpu(torch_dev, tmp_path):
    """the drop-in boundary end to end: a plain C program using only fftw3.h, linked with
    -lfftw3 from fftw3_amd/lib, executes on the GPU (host arrays are staged) and checks
    closed-form answers itself"""
    import os
    import subprocess
    from util import ROOT
    src = tmp_path / "client.c"
    exe = tmp_path / "client"
    src.write_text(C_CLIENT_RUN)
    libdir = os.path.join(ROOT, "fftw3_amd", "lib")
    subprocess.run(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir,
                    "-lfftw3", "-Wl,-rpath," + libdir, "-lm", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "client ok" in r.stdout, r.stdout


@pytest.mark.parametrize("n", [128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768])
def test_fused_real_rows_kernel(torch_dev, n):
    """r2c of contiguous rows of 128 ... 32768 reals runs as ONE fused step (pass + untangle; two register stages up
    to 2048, the three-stage rows kernel above):
    batched 1-D with a ragged last tile, and as the last dimension of 2-D / 3-D transforms"""
    from util import oracle_r2c, rrand
    torch, dev = torch_dev
    rng = np.random.default_rng(n)
    hm = 37
    x = rrand(rng, hm, n)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.zeros((hm, n // 2 + 1), dtype=torch.complex128, device=dev)
    p = fa.plan_many_dft_r2c(1, [n], hm, xd, None, 1, n, yd, None, 1, n // 2 + 1)
    assert "r2c-rows" in p.sprint() and "untangle" not in p.sprint(), p.sprint()
    p.execute()
    torch.cuda.synchronize()
    assert aerror(yd.cpu().numpy(), oracle_r2c(x, (n,), hm).reshape(hm, n // 2 + 1)) < TOL
    for shape in ((12, n), (3, 10, n)):
        size = int(np.prod(shape))
        hs = size // n * (n // 2 + 1)
        x = rrand(rng, 2, size)
        xd = torch.from_numpy(x).to(dev)
        yd = torch.zeros((2, hs), dtype=torch.complex128, device=dev)
        p = fa.plan_many_dft_r2c(len(shape), list(shape), 2, xd, None, 1, size, yd, None, 1, hs)
        assert "r2c-rows" in p.sprint(), p.sprint()
        p.execute()
        torch.cuda.synchronize()
        assert aerror(yd.cpu().numpy(), oracle_r2c(x, shape, 2).reshape(2, hs)) < TOL, shape
        # and back through the fused c2r rows kernel
        from util import oracle_c2r
        y = yd.cpu().numpy().reshape(-1).copy()
        zd = torch.zeros((2, size), dtype=torch.float64, device=dev)
        q = fa.plan_many_dft_c2r(len(shape), list(shape), 2, yd, None, 1, hs, zd, None, 1, size)
        assert "c2r-rows" in q.sprint(), q.sprint()
        q.execute()
        torch.cuda.synchronize()
        assert aerror(zd.cpu().numpy().reshape(-1), oracle_c2r(y, shape, 2)) < TOL, shape
    # batched 1-D c2r with a ragged last tile; the input is preserved
    y = oracle_r2c(rrand(rng, hm, n), (n,), hm)
    yd = torch.from_numpy(y).to(dev)
    zd = torch.zeros((hm, n), dtype=torch.float64, device=dev)
    q = fa.plan_many_dft_c2r(1, [n], hm, yd, None, 1, n // 2 + 1, zd, None, 1, n)
    assert "c2r-rows" in q.sprint() and "tangle" not in q.sprint(), q.sprint()
    q.execute()
    torch.cuda.synchronize()
    assert aerror(zd.cpu().numpy().reshape(-1), oracle_c2r(y, (n,), hm)) < TOL
    assert np.array_equal(yd.cpu().numpy(), y)


@pytest.mark.parametrize("shape", [(1024,), (20, 512), (6, 10, 256), (2000,), (12, 1000), (4096,), (3, 8192), (16384,), (2, 32768)])
def test_inplace_padded_real_transforms(torch_dev, shape):
    """FFTW's in-place real layout (rows padded to 2 (n/2 + 1) reals): r2c then c2r in the same
    buffer, through the fused rows kernels where they apply and the general path elsewhere"""
    from util import oracle_r2c, rrand
    torch, dev = torch_dev
    rng = np.random.default_rng(sum(shape))
    n = shape[-1]
    rows = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    hm, pad = 5, 2 * (n // 2 + 1)
    x = rrand(rng, hm, rows, n)
    buf = np.zeros((hm, rows, pad))
    buf[..., :n] = x
    bd = torch.from_numpy(buf).to(dev)
    cview = bd.view(-1).view(torch.complex128)
    size_r, size_c = rows * pad, rows * (pad // 2)
    p = fa.plan_many_dft_r2c(len(shape), list(shape), hm, bd, None, 1, size_r, cview, None, 1, size_c)
    p.execute()
    torch.cuda.synchronize()
    want = oracle_r2c(x, shape, hm)
    assert aerror(cview.cpu().numpy(), want) < TOL, p.sprint()
    q = fa.plan_many_dft_c2r(len(shape), list(shape), hm, cview, None, 1, size_c, bd, None, 1, size_r)
    q.execute()
    torch.cuda.synchronize()
    got = bd.cpu().numpy()[..., :n]
    assert aerror(got, x * np.prod(shape)) < TOL, q.sprint()


def _far_apart(n, rng=None, fill=None):
    """two independently allocated host arrays with at least one unrelated allocation between
    them (a guard block that must come back untouched)"""
    a = np.empty(n)
    guard = np.full(4096, 1234.5)
    b = np.empty(n)
    if rng is not None:
        a[:] = rng.random(n) - 0.5
        b[:] = rng.random(n) - 0.5
    elif fill is not None:
        a[:] = fill
        b[:] = fill
    return a, b, guard


def test_split_host_arrays_from_separate_allocations(torch_dev):
    """fftw_plan_guru_split_dft with ri / ii (ro / io) in different heap blocks, the usual way the
    split interface is called: the staging path must move the two planes separately (it used to copy
    ONE span from the lower to the upper plane -- everything in between, both ways).  c2c, r2c and c2r,
    through fftw_execute and the new-array split executes."""
    rng = np.random.default_rng(77)
    n, b = 1536, 3
    dims, hm = [(n, 1, 1)], [(b, n, n)]
    # ---- c2c
    ri, ii, g1 = _far_apart(n * b, rng)
    ro, io, g2 = _far_apart(n * b, fill=0.0)
    x = (ri + 1j * ii).reshape(b, n)
    want = oracle_dft(x, (n,), b).reshape(b, n)
    p = fa.plan_guru64_split_dft(dims, hm, ri, ii, ro, io)
    p.execute()
    assert aerror((ro + 1j * io).reshape(b, n), want) < TOL
    ri2, ii2, g3 = _far_apart(n * b, rng)
    ro2, io2, g4 = _far_apart(n * b, fill=0.0)
    p.execute_split_dft(ri2, ii2, ro2, io2)
    assert aerror((ro2 + 1j * io2).reshape(b, n), oracle_dft((ri2 + 1j * ii2).reshape(b, n), (n,), b).reshape(b, n)) < TOL
    # in place on the two planes
    ri3, ii3 = ri.copy(), ii.copy()
    q = fa.plan_guru64_split_dft(dims, hm, ri3, ii3, ri3, ii3)
    q.execute()
    assert aerror((ri3 + 1j * ii3).reshape(b, n), want) < TOL
    # ---- r2c: real input, split half spectrum (n/2+1 per row)
    nh = n // 2 + 1
    xr = rng.random((b, n)) - 0.5
    cro, cio, g5 = _far_apart(nh * b, fill=0.0)
    pr = fa.plan_guru64_split_dft_r2c([(n, 1, 1)], [(b, n, nh)], xr, cro, cio)
    pr.execute()
    wantr = oracle_r2c(xr, (n,), b).reshape(b, nh)
    assert aerror((cro + 1j * cio).reshape(b, nh), wantr) < TOL
    cro2, cio2, g6 = _far_apart(nh * b, fill=0.0)
    xr2 = rng.random((b, n)) - 0.5
    pr.execute_split_dft_r2c(xr2, cro2, cio2)
    assert aerror((cro2 + 1j * cio2).reshape(b, nh), oracle_r2c(xr2, (n,), b).reshape(b, nh)) < TOL
    # ---- c2r: split half spectrum in, real out (input is destroyed: work on copies)
    sre, sim_, g7 = _far_apart(nh * b)
    sre[:] = wantr.real.reshape(-1)
    sim_[:] = wantr.imag.reshape(-1)
    back = np.zeros((b, n))
    pc = fa.plan_guru64_split_dft_c2r([(n, 1, 1)], [(b, nh, n)], sre, sim_, back)
    pc.execute()
    assert aerror(back / n, xr) < TOL
    sre2, sim2, g8 = _far_apart(nh * b)
    sre2[:] = wantr.real.reshape(-1)
    sim2[:] = wantr.imag.reshape(-1)
    back2 = np.zeros((b, n))
    pc.execute_split_dft_c2r(sre2, sim2, back2)
    assert aerror(back2 / n, xr) < TOL
    for g in (g1, g2, g3, g4, g5, g6, g7, g8):
        assert np.all(g == 1234.5), "memory between the two planes was overwritten"


def test_unaligned_flag_and_new_array_execute_on_8_byte_offset(torch_dev):
    """FFTW_UNALIGNED (reference genus predicates, fftw/dft_simd/common/genus.c:26-331: a SIMD
    codelet may only run on aligned arrays; fftw_execute_dft needs the plan's alignment unless the
    plan was made with FFTW_UNALIGNED, A.c:433).  Here: a plan made with FFTW_UNALIGNED must run
    on arrays that are only 8-byte aligned, for every kernel family the planner would pick."""
    torch, dev = torch_dev
    rng = np.random.default_rng(5)
    for n, b in ((1024, 16), (4096, 4), (360, 9), (1 << 16, 2), (2000, 3), (97, 5)):
        x = crand(rng, b, n)
        want = oracle_dft(x, (n,), b).reshape(b, n)
        al = torch.zeros(b * n * 2 + 8, dtype=torch.float64, device=dev)
        ao = torch.zeros(b * n * 2 + 8, dtype=torch.float64, device=dev)
        p = fa.plan_many_dft(1, [n], b, al, None, 1, n, ao, None, 1, n, fa.FORWARD, fa.ESTIMATE | fa.UNALIGNED)
        # 8 bytes off a 16-byte boundary
        xin = al[1:1 + 2 * b * n]
        xout = ao[1:1 + 2 * b * n]
        assert xin.data_ptr() % 16 == 8 and xout.data_ptr() % 16 == 8
        xin.copy_(torch.from_numpy(x.view(np.float64).reshape(-1)))
        p.execute_dft(xin, xout)
        torch.cuda.synchronize()
        got = xout.cpu().numpy().view(np.complex128).reshape(b, n)
        assert aerror(got, want) < TOL, n
    # real transforms: the fused rows kernels have no unaligned form, so the flag must steer the plan
    n, b = 1024, 8
    xr = rrand(rng, b, n)
    al = torch.zeros(b * n + 8, dtype=torch.float64, device=dev)
    ao = torch.zeros(b * (n // 2 + 1) * 2 + 8, dtype=torch.float64, device=dev)
    p = fa.plan_many_dft_r2c(1, [n], b, al, None, 1, n, ao, None, 1, n // 2 + 1, fa.ESTIMATE | fa.UNALIGNED)
    xin = al[1:1 + b * n]
    xout = ao[1:1 + 2 * b * (n // 2 + 1)]
    xin.copy_(torch.from_numpy(xr.reshape(-1)))
    p.execute_dft_r2c(xin, xout)
    torch.cuda.synchronize()
    got = xout.cpu().numpy().view(np.complex128).reshape(b, n // 2 + 1)
    assert aerror(got, oracle_r2c(xr, (n,), b).reshape(b, n // 2 + 1)) < TOL


def test_pair_launches_of_the_two_pass_plan(torch_dev, monkeypatch):
    """batched N = 1024 x 1024 with more than one chunk, ONE lane (FFTW_AMD_LANES=1; round 3 made two chunk
    lanes the default): pass 2 of chunk c-1 and pass 1 of chunk c share a launch (fa_hip_launch_pair1024, two
    scratch slots).  Ragged last chunk, both signs, in place, new-array execute -- all against the oracle."""
    torch, dev = torch_dev
    monkeypatch.setenv("FFTW_AMD_LANES", "1")
    rng = np.random.default_rng(2020)
    n, b = 1 << 20, 5
    fa.set_chunk_bytes(32 << 20)                 # 2 transforms per chunk -> chunks of 2, 2, 1
    try:
        x = crand(rng, b, n)
        xd = torch.from_numpy(x).to(dev)
        for sign in (-1, 1):
            yd = torch.zeros_like(xd)
            p = fa.plan_many_dft(1, [n], b, xd, None, 1, n, yd, None, 1, n, sign)
            assert p.chunk == 2 and p.batch == 5
            p.execute()
            torch.cuda.synchronize()
            assert p.paired, "the pair launch did not engage"
            assert aerror(yd.cpu().numpy(), oracle_dft(x, (n,), b, sign).reshape(b, n)) < TOL
            # profiled execution reports every launch under step 0: chunks + 1 launches
            prof = p.execute_profiled()
            assert prof[0][2] == 4 and prof[1][2] == 0
            torch.cuda.synchronize()
            assert aerror(yd.cpu().numpy(), oracle_dft(x, (n,), b, sign).reshape(b, n)) < TOL
        # in place + new-array execute on other buffers
        zd = xd.clone()
        q = fa.plan_many_dft(1, [n], b, zd, None, 1, n, zd, None, 1, n, fa.FORWARD)
        q.execute()
        torch.cuda.synchronize()
        want = oracle_dft(x, (n,), b).reshape(b, n)
        assert q.paired and aerror(zd.cpu().numpy(), want) < TOL
        x2 = crand(rng, b, n)
        z2 = torch.from_numpy(x2).to(dev)
        q.execute_dft(z2, z2)
        torch.cuda.synchronize()
        assert aerror(z2.cpu().numpy(), oracle_dft(x2, (n,), b).reshape(b, n)) < TOL
    finally:
        fa.set_chunk_bytes(0)


@pytest.mark.parametrize("lanes", [2, 3, 4])
def test_chunk_lanes(torch_dev, monkeypatch, lanes):
    """chunk lanes (fa_run_locked, round 3): chunk c of a multi-pass plan runs all its steps on stream c % lanes
    in scratch slot c % lanes, the lanes overlap freely and join the caller's stream at the end.  Two-pass
    2^20 (ragged last chunk, both signs, in place, new-array execute, a second execution right behind the
    first on the same stream), a three-pass length, r2c with its untangle step, and a 2-D plan whose chunks
    hold several images -- all against the oracle."""
    torch, dev = torch_dev
    monkeypatch.setenv("FFTW_AMD_LANES", str(lanes))
    rng = np.random.default_rng(3000 + lanes)
    try:
        n, b = 1 << 20, 7
        fa.set_chunk_bytes(32 << 20)                 # chunks of 2, 2, 2, 1
        x = crand(rng, b, n)
        xd = torch.from_numpy(x).to(dev)
        for sign in (-1, 1):
            yd = torch.zeros_like(xd)
            p = fa.plan_many_dft(1, [n], b, xd, None, 1, n, yd, None, 1, n, sign)
            assert p.chunk == 2 and not p.paired
            p.execute()
            p.execute()                              # the second run must wait for the first one's lanes
            torch.cuda.synchronize()
            want = oracle_dft(x, (n,), b, sign).reshape(b, n)
            assert aerror(yd.cpu().numpy(), want) < TOL
            prof = p.execute_profiled()              # serialised lanes: 4 launches of each of the 2 steps
            assert [t[2] for t in prof] == [4, 4]
            assert aerror(yd.cpu().numpy(), want) < TOL
        zd = xd.clone()
        q = fa.plan_many_dft(1, [n], b, zd, None, 1, n, zd, None, 1, n, fa.FORWARD)
        q.execute()
        torch.cuda.synchronize()
        assert aerror(zd.cpu().numpy(), oracle_dft(x, (n,), b).reshape(b, n)) < TOL
        x2 = crand(rng, b, n)
        z2 = torch.from_numpy(x2).to(dev)
        q.execute_dft(z2, z2)
        torch.cuda.synchronize()
        assert aerror(z2.cpu().numpy(), oracle_dft(x2, (n,), b).reshape(b, n)) < TOL
        del xd, zd, z2, yd
        # three passes (2^23), r2c (passes + untangle), mixed radix
        fa.set_chunk_bytes(64 << 20)
        n, b = 1 << 23, 3
        x = crand(rng, b, n)
        y = gpu_c2c(torch_dev, x, (n,), b)
        assert aerror(y, oracle_dft(x, (n,), b).reshape(b, n)) < TOL
        n, b = 1 << 21, 5
        xr = rrand(rng, b, n)
        xd = torch.from_numpy(xr).to(dev)
        yd = torch.zeros(b, n // 2 + 1, dtype=torch.complex128, device=dev)
        p = fa.plan_many_dft_r2c(1, [n], b, xd, None, 1, n, yd, None, 1, n // 2 + 1)
        assert 1 <= p.chunk < b, p.sprint()
        p.execute()
        torch.cuda.synchronize()
        assert aerror(yd.cpu().numpy(), oracle_r2c(xr, (n,), b).reshape(b, n // 2 + 1)) < TOL
        fa.set_chunk_bytes(8 << 20)
        n0, n1, b = 300, 200, 9                      # 2-D, several images per chunk
        x = crand(rng, b, n0 * n1)
        y = gpu_c2c(torch_dev, x, (n0, n1), b)
        assert aerror(y, oracle_dft(x, (n0, n1), b).reshape(b, n0 * n1)) < TOL
    finally:
        fa.set_chunk_bytes(0)


def test_new_array_execute_on_less_aligned_arrays_falls_back(torch_dev):
    """A plan made WITHOUT FFTW_UNALIGNED on 16-byte aligned arrays may hold one-trip rows steps that have
    no executor for other alignments (8192-point rows, fused real rows, one-kernel Bluestein).  Executing it
    on arrays 8 bytes off (FFTW calls that a caller error, fftw3.h / A.c:433-440) used to abort() the
    process; now fa_run switches to the plan's FFTW_UNALIGNED twin.  Results against the oracle, and the
    aligned path keeps its one-trip step."""
    torch, dev = torch_dev
    rng = np.random.default_rng(77)
    for n, b in ((8192, 6), (16384, 3), (1031, 5), (5000, 4)):
        x = crand(rng, b, n)
        want = oracle_dft(x, (n,), b).reshape(b, n)
        al = torch.zeros(b * n * 2 + 8, dtype=torch.float64, device=dev)
        ao = torch.zeros(b * n * 2 + 8, dtype=torch.float64, device=dev)
        p = fa.plan_many_dft(1, [n], b, al, None, 1, n, ao, None, 1, n, fa.FORWARD)
        assert len(p.steps()) == 1, p.sprint()
        xin, xout = al[1:1 + 2 * b * n], ao[1:1 + 2 * b * n]
        assert xin.data_ptr() % 16 == 8 and xout.data_ptr() % 16 == 8
        xin.copy_(torch.from_numpy(x.view(np.float64).reshape(-1)))
        p.execute_dft(xin, xout)
        torch.cuda.synchronize()
        assert aerror(xout.cpu().numpy().view(np.complex128).reshape(b, n), want) < TOL, n
        # and the plan's own arrays still take the one-trip step
        al[:2 * b * n].copy_(torch.from_numpy(x.view(np.float64).reshape(-1)))
        p.execute()
        torch.cuda.synchronize()
        assert aerror(ao[:2 * b * n].cpu().numpy().view(np.complex128).reshape(b, n), want) < TOL, n
    # fused real rows
    n, b = 4096, 8
    xr = rrand(rng, b, n)
    al = torch.zeros(b * n + 8, dtype=torch.float64, device=dev)
    ao = torch.zeros(b * (n // 2 + 1) * 2 + 8, dtype=torch.float64, device=dev)
    p = fa.plan_many_dft_r2c(1, [n], b, al, None, 1, n, ao, None, 1, n // 2 + 1)
    assert len(p.steps()) == 1, p.sprint()
    xin = al[1:1 + b * n]
    xout = ao[1:1 + 2 * b * (n // 2 + 1)]
    xin.copy_(torch.from_numpy(xr.reshape(-1)))
    p.execute_dft_r2c(xin, xout)
    torch.cuda.synchronize()
    got = xout.cpu().numpy().view(np.complex128).reshape(b, n // 2 + 1)
    assert aerror(got, oracle_r2c(xr, (n,), b).reshape(b, n // 2 + 1)) < TOL


def test_host_arrays_are_staged_chunk_by_chunk(torch_dev):
    """dense batched transforms on plain host arrays that run in several chunks: upload, compute and download
    of different chunks overlap on separate streams (fa_run_locked, host pipeline).  Ordinary launches (2-pass
    n = 65536), pair launches (n = 2^20), r2c, ragged last chunk; results against the oracle, input untouched."""
    rng = np.random.default_rng(99)
    try:
        for n, b, chunk_bytes in ((1 << 16, 11, 2 << 20), (1 << 20, 5, 32 << 20)):
            fa.set_chunk_bytes(chunk_bytes)
            x = crand(rng, b, n)
            x0 = x.copy()
            y = np.zeros_like(x)
            p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, fa.FORWARD)
            assert 1 < p.chunk < b
            p.execute()
            assert np.array_equal(x, x0)
            assert aerror(y, oracle_dft(x, (n,), b).reshape(b, n)) < TOL, n
            # new arrays, again (streams and staging buffers are reused)
            x2 = crand(rng, b, n)
            y2 = np.zeros_like(x2)
            p.execute_dft(x2, y2)
            assert aerror(y2, oracle_dft(x2, (n,), b).reshape(b, n)) < TOL, n
        n, b = 1 << 17, 7
        fa.set_chunk_bytes(4 << 20)
        xr = rrand(rng, b, n)
        yr = np.zeros((b, n // 2 + 1), dtype=np.complex128)
        p = fa.plan_many_dft_r2c(1, [n], b, xr, None, 1, n, yr, None, 1, n // 2 + 1)
        assert 1 <= p.chunk < b
        p.execute()
        assert aerror(yr, oracle_r2c(xr, (n,), b).reshape(b, n // 2 + 1)) < TOL
    finally:
        fa.set_chunk_bytes(0)


def test_planner_returns_null_when_the_device_is_out_of_memory(torch_dev):
    """tables and scratch are allocated at plan time: when the device cannot provide them the planner returns
    NULL (as every fftw_plan_* call may, fftw3.h) instead of aborting, and nothing leaks"""
    torch, dev = torch_dev
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    hog = torch.empty(max(0, free0 - (160 << 20)), dtype=torch.uint8, device=dev)     # leave ~160 MiB
    fa.set_chunk_bytes(1 << 30)
    try:
        n, b = 1 << 20, 256
        x = torch.zeros(8, dtype=torch.complex128, device=dev)      # the planner only needs addresses
        with pytest.raises(ValueError):
            fa.plan_many_dft(1, [n], b, x, None, 1, n, x, None, 1, n, fa.FORWARD)      # needs 2 lanes x 1 GiB of scratch
        free1, _ = torch.cuda.mem_get_info()
        assert free1 > (96 << 20)                                   # the failed plan gave its tables back
    finally:
        fa.set_chunk_bytes(0)
        del hog
        torch.cuda.empty_cache()
    # and planning works again afterwards
    rng = np.random.default_rng(4)
    xs = crand(rng, 2, 4096)
    assert aerror(gpu_c2c(torch_dev, xs, (4096,), 2), oracle_dft(xs, (4096,), 2).reshape(2, 4096)) < TOL
