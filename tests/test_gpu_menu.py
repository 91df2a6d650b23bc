"""Every two-stage register kernel of fftw3_amd/csrc/rr_menu.inc, in every lane-mapping /
twiddle variant the planner can emit, against the oracle:
  (L,L,0)  contiguous single pass            n = L, howmany = 300
  (T,T,0)  single column pass                n = L, interleaved batch (stride = howmany)
  (T,T,0) + (L,T,2)  two-pass plan           n = L * L
  (T,T,1) x2 and (L,T,0)                     forced splits L x L x 8 (or 64) and 64 x 8 x L
and that the plan really uses the register kernel for that length."""
import os
import re

import numpy as np
import pytest

import fftw3_amd as fa
from util import ROOT, TOL, aerror, crand, oracle_dft

pytestmark = pytest.mark.gpu


def menu():
    out = []
    with open(os.path.join(ROOT, "fftw3_amd", "csrc", "rr_menu.inc")) as f:
        for m in re.finditer(r"X\((\d+), (\d+), (\d+)\)", f.read()):
            out.append(tuple(int(v) for v in m.groups()))
    return out


MENU = menu()


def menu3():
    out = []
    with open(os.path.join(ROOT, "fftw3_amd", "csrc", "r3_menu.inc")) as f:
        for m in re.finditer(r"X\((\d+), (\d+), (\d+), (\d+)\)", f.read()):
            out.append(tuple(int(v) for v in m.groups()))
    return out


MENU3 = menu3()


def menu3t():
    out = []
    with open(os.path.join(ROOT, "fftw3_amd", "csrc", "r3t_menu.inc")) as f:
        for m in re.finditer(r"X\((\d+), (\d+), (\d+), (\d+)\)", f.read()):
            out.append(tuple(int(v) for v in m.groups()))
    return out


MENU3T = menu3t()


def _run(n, hm, stride, dist):
    import torch
    rng = np.random.default_rng(n + hm)
    x = crand(rng, hm * n)
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros_like(xd)
    p = fa.plan_many_dft(1, [n], hm, xd, None, stride, dist, yd, None, stride, dist, fa.FORWARD)
    p.execute()
    p.sync()
    want = np.zeros_like(x)
    oracle_dft(x, (n,), hm, out=want, istride=stride, idist=dist, ostride=stride, odist=dist)
    return p, aerror(yd.cpu().numpy(), want)


def test_menu_is_complete_and_consistent():
    assert len(MENU) >= 60
    for L, r1, r2 in MENU:
        assert r1 * r2 == L


@pytest.mark.parametrize("L,r1,r2", MENU, ids=[str(m[0]) for m in MENU])
def test_single_pass_rows_and_columns(L, r1, r2, monkeypatch):
    # contiguous rows of a length that also has a three-stage rows kernel (525 ... 648) may take that one
    rows3 = any(m[0] == L for m in MENU3)
    if L <= 32:                                    # dense rows of 16 ... 32 points would take the one-stage kernel
        monkeypatch.setenv("FFTW_AMD_NO_R1", "1")
    p, e = _run(L, 300, 1, L)
    assert "pass-%d/reg2" % L in p.sprint() or (rows3 and "pass-%d/reg3" % L in p.sprint()), p.sprint()
    assert e <= TOL, (L, e)
    p, e = _run(L, 300, 300, 1)
    assert "pass-%d/reg2" % L in p.sprint() or "pass-%d/reg3" % L in p.sprint(), p.sprint()
    assert e <= TOL, (L, e)


@pytest.mark.parametrize("L,r1,r2", MENU, ids=[str(m[0]) for m in MENU])
def test_two_pass_square(L, r1, r2, monkeypatch):
    # the planner's cost model may prefer another split of L * L: this test is about the kernel of length L
    monkeypatch.setenv("FFTW_AMD_FORCE_LENS", "%d,%d" % (L, L))
    p, e = _run(L * L, 3, 1, L * L)
    s = p.sprint()
    assert s.count("pass-%d/reg2" % L) == 2 or L * L <= 4096, s
    assert e <= TOL, (L, e)


@pytest.mark.parametrize("L,r1,r2", MENU, ids=[str(m[0]) for m in MENU])
def test_three_pass_output_twiddle_and_last_pass_variants(L, r1, r2, monkeypatch):
    """(T,T,1): first two passes of a forced L x L x 8 split; (L,T,0): last pass of 64 x 8 x L"""
    m = 8 if L > 200 else 64          # the middle pass needs enough sequences to fill a tile
    monkeypatch.setenv("FFTW_AMD_FORCE_LENS", "%d,%d,%d" % (L, L, m))
    p, e = _run(L * L * m, 1, 1, L * L * m)
    assert p.sprint().count("pass-%d/reg2" % L) >= 2, p.sprint()
    assert e <= TOL, (L, e)
    monkeypatch.setenv("FFTW_AMD_FORCE_LENS", "64,8,%d" % L)
    p, e = _run(512 * L, 4, 1, 512 * L)
    assert "pass-%d/reg2" % L in p.sprint(), p.sprint()
    assert e <= TOL, (L, e)


@pytest.mark.parametrize("L,r1,r2,r3", MENU3, ids=[str(m[0]) for m in MENU3])
def test_three_stage_rows_kernel(L, r1, r2, r3):
    """every three-stage rows kernel of r3_menu.inc: forward out of place with a ragged last
    tile, backward in place (swap flags), against the oracle"""
    import torch
    assert r1 * r2 * r3 == L
    hm = 37
    p, e = _run(L, hm, 1, L)
    assert "pass-%d/reg3" % L in p.sprint(), p.sprint()
    assert e <= TOL, (L, e)
    rng = np.random.default_rng(L)
    x = crand(rng, hm * L)
    xd = torch.from_numpy(x).cuda()
    q = fa.plan_many_dft(1, [L], hm, xd, None, 1, L, xd, None, 1, L, fa.BACKWARD)
    assert "pass-%d/reg3" % L in q.sprint()
    q.execute()
    q.sync()
    assert aerror(xd.cpu().numpy(), oracle_dft(x, (L,), hm, sign=+1)) <= TOL, L


def menu3w():
    out = []
    with open(os.path.join(ROOT, "fftw3_amd", "csrc", "r3w_menu.inc")) as f:
        for m in re.finditer(r"X\((\d+), (\d+), (\d+), (\d+)\)", f.read()):
            out.append(tuple(int(v) for v in m.groups()))
    return out


MENU3W = menu3w()


@pytest.mark.parametrize("L,r1,r2,r3", MENU3W, ids=[str(m[0]) for m in MENU3W])
def test_wide_three_stage_rows_kernel(L, r1, r2, r3):
    """every wide three-stage rows kernel of r3w_menu.inc (round 3: rows of 8193 ... 16383 points, one row per
    workgroup of 512 work-items, kernels_r3w.hip): forward out of place, backward in place (swap flags), against
    the oracle; the plan is ONE step"""
    import torch
    assert r1 * r2 * r3 == L and 8192 < L < 16384
    hm = 5
    p, e = _run(L, hm, 1, L)
    assert "pass-%d/reg3" % L in p.sprint() and len(p.steps()) == 1, p.sprint()
    assert e <= TOL, (L, e)
    rng = np.random.default_rng(L)
    x = crand(rng, hm * L)
    xd = torch.from_numpy(x).cuda()
    q = fa.plan_many_dft(1, [L], hm, xd, None, 1, L, xd, None, 1, L, fa.BACKWARD)
    assert len(q.steps()) == 1, q.sprint()
    q.execute()
    q.sync()
    assert aerror(xd.cpu().numpy(), oracle_dft(x, (L,), hm, sign=+1)) <= TOL, L


@pytest.mark.parametrize("L,r1,r2,r3", MENU3T, ids=[str(m[0]) for m in MENU3T])
def test_three_stage_strided_forms(L, r1, r2, r3, monkeypatch):
    """every strided three-stage kernel of r3t_menu.inc in the four forms the planner emits:
    (T,T,0) single column pass and first pass of L x L, (L,T,2) its second pass,
    (T,T,1) x2 in a forced L x L x 8 split, (L,T,0) as the last pass of 64 x 8 x L"""
    assert r1 * r2 * r3 == L
    if L > 1024:
        # 4-wide tiles (64-byte segments): the planner uses this kernel for 2^21 = 2048 x 1024 only; reach
        # the column form, the transposed-last form with input twiddle and the output-twiddle form by force
        monkeypatch.setenv("FFTW_AMD_FORCE_LENS", "%d,1024" % L)
        p, e = _run(L * 1024, 2, 1, L * 1024)
        assert "pass-%d/reg3" % L in p.sprint(), p.sprint()
        assert e <= TOL, (L, e)
        monkeypatch.setenv("FFTW_AMD_FORCE_LENS", "1024,%d" % L)
        p, e = _run(L * 1024, 2, 1, L * 1024)
        assert "pass-%d/reg3" % L in p.sprint(), p.sprint()
        assert e <= TOL, (L, e)
        monkeypatch.setenv("FFTW_AMD_FORCE_LENS", "%d,%d,8" % (L, 64))
        p, e = _run(L * 512, 2, 1, L * 512)
        assert "pass-%d/reg3" % L in p.sprint(), p.sprint()
        assert e <= TOL, (L, e)
        return
    p, e = _run(L, 300, 300, 1)                       # interleaved batch: one column pass
    assert "pass-%d/reg3" % L in p.sprint(), p.sprint()
    assert e <= TOL, (L, e)
    monkeypatch.setenv("FFTW_AMD_FORCE_LENS", "%d,%d" % (L, L))
    p, e = _run(L * L, 3, 1, L * L)
    assert p.sprint().count("pass-%d/reg3" % L) == 2, p.sprint()
    assert e <= TOL, (L, e)
    monkeypatch.setenv("FFTW_AMD_FORCE_LENS", "%d,%d,8" % (L, L))
    p, e = _run(L * L * 8, 1, 1, L * L * 8)
    assert p.sprint().count("pass-%d/reg3" % L) == 2, p.sprint()
    assert e <= TOL, (L, e)
    monkeypatch.setenv("FFTW_AMD_FORCE_LENS", "64,8,%d" % L)
    p, e = _run(512 * L, 4, 1, 512 * L)
    assert "pass-%d/reg3" % L in p.sprint(), p.sprint()
    assert e <= TOL, (L, e)


@pytest.mark.parametrize("L", list(range(2, 33)))
def test_one_stage_rows_kernel(L, monkeypatch):
    """dense rows of 2 ... 32 points, one butterfly per row (pass1r.hpp): a batch that ends inside a tile and one that
    ends inside the first 256-element run of a tile, forward out of place and backward in place, against the oracle;
    rows that are not dense (idist > n) and the FFTW_AMD_NO_R1 hook take the other kernels and agree"""
    import torch
    tile = 256 * (6 if L == 4 else min(8, 32 // L))
    for hm in (2 * tile + 257, tile + 3):
        p, e = _run(L, hm, 1, L)
        assert "pass-%d/reg1" % L in p.sprint(), p.sprint()
        assert e <= TOL, (L, hm, e)
    hm = tile + 300
    rng = np.random.default_rng(L)
    x = crand(rng, hm * L)
    xd = torch.from_numpy(x).cuda()
    q = fa.plan_many_dft(1, [L], hm, xd, None, 1, L, xd, None, 1, L, fa.BACKWARD)
    assert "pass-%d/reg1" % L in q.sprint()
    q.execute()
    q.sync()
    assert aerror(xd.cpu().numpy(), oracle_dft(x, (L,), hm, sign=+1)) <= TOL, L
    x = crand(rng, 300 * (L + 1))
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros_like(xd)
    p = fa.plan_many_dft(1, [L], 300, xd, None, 1, L + 1, yd, None, 1, L + 1, fa.FORWARD)
    assert "reg1" not in p.sprint(), p.sprint()
    p.execute()
    p.sync()
    want = oracle_dft(np.ascontiguousarray(x.reshape(300, L + 1)[:, :L]), (L,), 300).reshape(300, L)
    got = yd.cpu().numpy().reshape(300, L + 1)
    assert aerror(got[:, :L], want) <= TOL and np.all(got[:, L] == 0), L
    monkeypatch.setenv("FFTW_AMD_NO_R1", "1")
    p, e = _run(L, hm, 1, L)
    assert "reg1" not in p.sprint() and e <= TOL, (L, e)


def menu3r():
    out = []
    with open(os.path.join(ROOT, "fftw3_amd", "csrc", "r3r_menu.inc")) as f:
        for m in re.finditer(r"X\((\d+), (\d+), (\d+), (\d+)\)", f.read()):
            out.append(tuple(int(v) for v in m.groups()))
    return out


MENU3R = menu3r()


def menu3rw():
    out = []
    with open(os.path.join(ROOT, "fftw3_amd", "csrc", "r3rw_menu.inc")) as f:
        for m in re.finditer(r"X\((\d+), (\d+), (\d+), (\d+)\)", f.read()):
            out.append(tuple(int(v) for v in m.groups()))
    return out


MENU3RW = menu3rw()        # round 3: half lengths 9000 ... 15360 on the 512-item kernels (kernels_r3w.hip)


def menu2r():
    out = []
    with open(os.path.join(ROOT, "fftw3_amd", "csrc", "r2cr_menu.inc")) as f:
        for m in re.finditer(r"X\((\d+), (\d+), (\d+)\)", f.read()):
            out.append(tuple(int(v) for v in m.groups()) + (1,))
    return out


MENU2R = menu2r()


@pytest.mark.parametrize("L,r1,r2,r3", MENU2R + MENU3R + MENU3RW + [(2048, 8, 16, 16), (4096, 16, 16, 16), (8192, 32, 16, 16), (16384, 32, 16, 32)],
                         ids=[str(m[0]) for m in MENU2R + MENU3R + MENU3RW] + ["2048", "4096", "8192", "16384"])
def test_mixed_and_three_stage_real_rows(L, r1, r2, r3):
    """real rows of n = 2L in one trip -- the mixed-radix two-stage lengths of r2cr_menu.inc (r2crows.hpp) and the
    three-stage lengths (pass3g / pass3s MODE 1, 2): r2c with a ragged last tile against the oracle, c2r of the result
    (out of place: input preserved; then in FFTW's padded in-place layout) against n x"""
    import torch
    from util import oracle_r2c, oracle_c2r, rrand
    assert r1 * r2 * r3 == L
    n, hm = 2 * L, (11 if L > 648 else 4096 // L * 2 + 3)
    rng = np.random.default_rng(L)
    x = rrand(rng, hm, n)
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros((hm, L + 1), dtype=torch.complex128, device="cuda")
    p = fa.plan_many_dft_r2c(1, [n], hm, xd, None, 1, n, yd, None, 1, L + 1)
    assert "pass-%d/r2c-rows" % L in p.sprint() and len(p.steps()) == 1, p.sprint()
    p.execute()
    p.sync()
    want = oracle_r2c(x, (n,), hm).reshape(hm, L + 1)
    assert aerror(yd.cpu().numpy(), want) <= TOL, L
    y = yd.cpu().numpy().copy()
    zd = torch.zeros((hm, n), dtype=torch.float64, device="cuda")
    q = fa.plan_many_dft_c2r(1, [n], hm, yd, None, 1, L + 1, zd, None, 1, n)
    assert "pass-%d/c2r-rows" % L in q.sprint() and len(q.steps()) == 1, q.sprint()
    q.execute()
    q.sync()
    assert aerror(zd.cpu().numpy().reshape(-1), oracle_c2r(want.reshape(-1), (n,), hm)) <= TOL, L
    assert np.array_equal(yd.cpu().numpy(), y)
    # in place, rows padded to 2 (L + 1) reals
    buf = np.zeros((hm, 2 * (L + 1)))
    buf[:, :n] = x
    bd = torch.from_numpy(buf).cuda()
    cv = bd.view(-1).view(torch.complex128)
    pi = fa.plan_many_dft_r2c(1, [n], hm, bd, None, 1, 2 * (L + 1), cv, None, 1, L + 1)
    assert len(pi.steps()) == 1, pi.sprint()
    pi.execute()
    pi.sync()
    assert aerror(cv.cpu().numpy().reshape(hm, L + 1), want) <= TOL, L
    qi = fa.plan_many_dft_c2r(1, [n], hm, cv, None, 1, L + 1, bd, None, 1, 2 * (L + 1))
    assert len(qi.steps()) == 1, qi.sprint()
    qi.execute()
    qi.sync()
    assert aerror(bd.cpu().numpy()[:, :n], x * n) <= TOL, L


def blue_menu():
    out = []
    for name in ("blue_menu.inc", "bluew_menu.inc"):          # 256 / 512 work-items per workgroup
        with open(os.path.join(ROOT, "fftw3_amd", "csrc", name)) as f:
            for m in re.finditer(r"X\((\d+), (\d+), (\d+), (\d+)\)", f.read()):
                out.append(int(m.group(1)))
    return out


BLUE = blue_menu()


def _largest_fitting_hard_length(nb, lo):
    """largest n with 2n - 1 <= nb, above lo, whose largest prime factor exceeds 31 (so the planner needs Bluestein)"""
    def lpf(v):
        p, best = 2, 1
        while p * p <= v:
            while v % p == 0:
                best, v = p, v // p
            p += 1
        return max(best, v) if v > 1 else best
    n = (nb + 1) // 2
    while n > lo and lpf(n) <= 31:
        n -= 1
    return n if n > lo else 0


@pytest.mark.parametrize("idx", list(range(len(BLUE))), ids=[str(v) for v in BLUE])
def test_bluestein_rows_kernel(idx, monkeypatch):
    """every padded length of blue_menu.inc / bluew_menu.inc: rows of the largest length that needs it, forward out of place with a
    ragged last tile and backward in place, against the oracle; the step-by-step plan (FFTW_AMD_NO_BLUE_ROWS) agrees"""
    import torch
    nb = BLUE[idx]
    n = _largest_fitting_hard_length(nb, (BLUE[idx - 1] + 1) // 2 if idx else 200)
    if not n:
        pytest.skip("no length needs this padded size")
    hm = 2 * max(1, 8192 // nb) + 1
    p, e = _run(n, hm, 1, n)
    assert "pass-%d/bluestein-rows n=%d" % (nb, n) in p.sprint() and len(p.steps()) == 1, p.sprint()
    assert e <= TOL, (n, e)
    rng = np.random.default_rng(n)
    x = crand(rng, hm * n)
    xd = torch.from_numpy(x).cuda()
    q = fa.plan_many_dft(1, [n], hm, xd, None, 1, n, xd, None, 1, n, fa.BACKWARD)
    assert "bluestein-rows" in q.sprint(), q.sprint()
    q.execute()
    q.sync()
    assert aerror(xd.cpu().numpy(), oracle_dft(x, (n,), hm, sign=+1)) <= TOL, n
    monkeypatch.setenv("FFTW_AMD_NO_BLUE_ROWS", "1")
    p, e = _run(n, hm, 1, n)
    assert "bluestein-rows" not in p.sprint() and e <= TOL, (n, e)


@pytest.mark.parametrize("L", list(range(2, 33)))
def test_one_stage_real_rows_kernel(L, monkeypatch):
    """dense real rows of n = 2L = 4 ... 64 points in one trip, one work-item per row (pass1r_real_kernel): r2c with
    batches that end inside a tile / inside the first run of a tile against the oracle, c2r of the result against n x
    with its input preserved; the padded in-place layout (rows not dense) and FFTW_AMD_NO_R1 take the other kernels"""
    import torch
    from util import oracle_r2c, oracle_c2r, rrand
    n = 2 * L
    tile = 256 * (6 if L == 4 else min(8, 32 // L))
    rng = np.random.default_rng(100 + L)
    for hm in (2 * tile + 257, tile + 3):
        x = rrand(rng, hm, n)
        xd = torch.from_numpy(x).cuda()
        yd = torch.zeros((hm, L + 1), dtype=torch.complex128, device="cuda")
        p = fa.plan_many_dft_r2c(1, [n], hm, xd, None, 1, n, yd, None, 1, L + 1)
        assert "pass-%d/r2c-rows" % L in p.sprint() and len(p.steps()) == 1, p.sprint()
        p.execute()
        p.sync()
        want = oracle_r2c(x, (n,), hm).reshape(hm, L + 1)
        assert aerror(yd.cpu().numpy(), want) <= TOL, (L, hm)
        y = yd.cpu().numpy().copy()
        zd = torch.zeros((hm, n), dtype=torch.float64, device="cuda")
        q = fa.plan_many_dft_c2r(1, [n], hm, yd, None, 1, L + 1, zd, None, 1, n)
        assert "pass-%d/c2r-rows" % L in q.sprint() and len(q.steps()) == 1, q.sprint()
        q.execute()
        q.sync()
        assert aerror(zd.cpu().numpy(), x * n) <= TOL, (L, hm)
        assert np.array_equal(yd.cpu().numpy(), y)
    hm = 700
    x = rrand(rng, hm, n)
    buf = np.zeros((hm, 2 * (L + 1)))
    buf[:, :n] = x
    bd = torch.from_numpy(buf).cuda()
    cv = bd.view(-1).view(torch.complex128)
    pi = fa.plan_many_dft_r2c(1, [n], hm, bd, None, 1, 2 * (L + 1), cv, None, 1, L + 1)
    assert "pass-%d/r2c-rows" % L in pi.sprint() and len(pi.steps()) == 1, pi.sprint()   # padded rows: the PAD form
    pi.execute()
    pi.sync()
    want = oracle_r2c(x, (n,), hm).reshape(hm, L + 1)
    assert aerror(cv.cpu().numpy().reshape(hm, L + 1), want) <= TOL, L
    qi = fa.plan_many_dft_c2r(1, [n], hm, cv, None, 1, L + 1, bd, None, 1, 2 * (L + 1))
    assert "pass-%d/c2r-rows" % L in qi.sprint() and len(qi.steps()) == 1, qi.sprint()
    qi.execute()
    qi.sync()
    assert aerror(bd.cpu().numpy()[:, :n], x * n) <= TOL, L
    # out of place into padded real rows: the padding is not written
    yd = torch.from_numpy(want.copy()).cuda()
    od = torch.full((hm, 2 * (L + 1)), 7.5, dtype=torch.float64, device="cuda")
    qo = fa.plan_many_dft_c2r(1, [n], hm, yd, None, 1, L + 1, od, None, 1, 2 * (L + 1))
    assert "pass-%d/c2r-rows" % L in qo.sprint(), qo.sprint()
    qo.execute()
    qo.sync()
    o = od.cpu().numpy()
    assert aerror(o[:, :n], x * n) <= TOL and np.all(o[:, n:] == 7.5), L
    monkeypatch.setenv("FFTW_AMD_NO_R1", "1")
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros((hm, L + 1), dtype=torch.complex128, device="cuda")
    p = fa.plan_many_dft_r2c(1, [n], hm, xd, None, 1, n, yd, None, 1, L + 1)
    assert "r2c-rows" not in p.sprint(), p.sprint()
    p.execute()
    p.sync()
    assert aerror(yd.cpu().numpy(), oracle_r2c(x, (n,), hm).reshape(hm, L + 1)) <= TOL, L
