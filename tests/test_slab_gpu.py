"""fftw3_amd.slab on one GPU (world = 1: the exchange is a device copy): the local
plans, strides and copy steps of the distributed pipeline run on the HIP path.
The multi-rank exchange itself is covered by the gloo tests in test_distributed.py."""
import numpy as np
import pytest

import fftw3_amd as fa
from fftw3_amd import slab
from util import TOL, aerror, oracle_dft, oracle_r2c, oracle_r2r

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_slab_gpu_c2c_vs_oracle():
    import torch
    rng = np.random.default_rng(3)
    for n, hm, flags in [([48, 64], 1, 0), ([30, 20, 12], 2, slab.TRANSPOSED_OUT), ([64, 9], 3, slab.TRANSPOSED_IN)]:
        tot = int(np.prod(n)) * hm
        full = ((rng.random(tot) - 0.5) + 1j * (rng.random(tot) - 0.5)).reshape(n + [hm])
        src = np.swapaxes(full, 0, 1) if flags & slab.TRANSPOSED_IN else full
        x = _dev(src.reshape(-1))
        y = torch.zeros_like(x)
        p = slab.plan_many_dft(n, hm, 0, 0, x, y, fa.FORWARD, fa.ESTIMATE | flags, world=1, rank=0)
        p.execute()
        p.sync()
        want = np.empty_like(full)
        for h in range(hm):
            want[..., h] = oracle_dft(np.ascontiguousarray(full[..., h]).reshape(-1), tuple(n), 1).reshape(n)
        if flags & slab.TRANSPOSED_OUT:
            want = np.swapaxes(want, 0, 1)
        assert aerror(y.cpu().numpy(), np.ascontiguousarray(want).reshape(-1)) <= TOL


def test_slab_gpu_real_and_r2r():
    import torch
    rng = np.random.default_rng(4)
    n = [40, 18, 10]
    nh = n[-1] // 2 + 1
    full = rng.random(n) - 0.5
    pad = np.zeros(n[:-1] + [2 * nh])
    pad[..., :n[-1]] = full
    xr = _dev(pad.reshape(-1))
    yc = torch.zeros(n[0] * n[1] * nh, dtype=torch.complex128, device="cuda")
    p = slab.plan_dft_r2c_3d(n[0], n[1], n[2], xr, yc, world=1, rank=0)
    p.execute()
    p.sync()
    assert aerror(yc.cpu().numpy(), oracle_r2c(full.reshape(-1), tuple(n), 1)) <= TOL
    zr = torch.zeros_like(xr)
    q = slab.plan_dft_c2r_3d(n[0], n[1], n[2], yc, zr, world=1, rank=0)
    q.execute()
    q.sync()
    got = zr.cpu().numpy().reshape(n[:-1] + [2 * nh])[..., :n[-1]]
    assert aerror(got, full * np.prod(n)) <= TOL
    x = _dev(full.reshape(-1))
    y = torch.zeros_like(x)
    kinds = [fa.REDFT10, fa.RODFT00, fa.DHT]
    r = slab.plan_r2r_3d(n[0], n[1], n[2], x, y, *kinds, world=1, rank=0)
    r.execute()
    r.sync()
    assert aerror(y.cpu().numpy(), oracle_r2r(full.reshape(-1), n, kinds)) <= TOL


def test_slab_gpu_large_2d_equals_single_plan():
    """4096 x 4096: the slab pipeline against the ordinary fftw_plan_dft_2d of the same library"""
    import torch
    n0 = n1 = 4096
    x = torch.view_as_complex(torch.rand((n0 * n1, 2), dtype=torch.float64, device="cuda") - 0.5)
    y = torch.zeros_like(x)
    z = torch.zeros_like(x)
    p = slab.plan_dft_2d(n0, n1, x, y, fa.FORWARD, world=1, rank=0)
    q = fa.plan_dft_2d(n0, n1, x, z, fa.FORWARD)
    p.execute()
    q.execute()
    q.sync()
    p.sync()
    err = (y - z).abs().max().item() / z.abs().max().item()
    assert err <= TOL


def test_slab_gpu_two_ranks_share_the_gpu_gloo_exchange():
    """two processes, both on cuda:0, gloo all-to-all on device tensors: the multi-rank pipeline
    (uneven blocks, all kinds, TRANSPOSED flags) with every local plan on the HIP path"""
    import os
    from test_distributed import _run_ranks
    from util import ROOT
    os.environ["FA_SLAB_DEVICE"] = "cuda"
    try:
        _run_ranks(os.path.join(ROOT, "tests", "workers", "slab_worker.py"), 2, timeout=600)
    finally:
        del os.environ["FA_SLAB_DEVICE"]
