"""The slab-decomposed transform behind the C ABI (include/fftw3_amd.h, fftw3_amd/csrc/slab.c): the c2c part of
the reference's distributed-memory API (fftw/mpi/fftw3-mpi.h:74-215) with the communicator replaced by a list of
devices of one process.  The one-GPU box runs it with the same device named several times (devs = {0, 0, ...}): the
local plans, the peer-to-peer exchanges and the event ordering are the same, only the copies stay on one card."""
import numpy as np
import pytest

import fftw3_amd as fa
from fftw3_amd import slab
from util import TOL, aerror, crand, oracle_dft


def test_local_size_matches_the_reference_block_rule():
    for n in ([7, 5], [64, 64], [100, 3], [5, 4, 3], [4096, 4096]):
        for ndev in (1, 2, 3, 8):
            cover = 0
            for g in range(ndev):
                tot, ln0, lo = fa.slab_local_size(n, ndev, g)
                want = slab.local_size(n, ndev, g)                    # (alloc, local_n0, local_0_start) of fftw3-mpi
                assert (ln0, lo) == (want[1], want[2])
                assert tot == ln0 * int(np.prod(n[1:]))
                assert lo == cover or ln0 == 0
                cover += ln0
            assert cover == n[0]


def test_slab_plan_is_built_without_a_device():
    """planning needs no GPU: the local plans are ordinary plans of the library (rows over the trailing dimensions,
    columns down the first one at the stride of the device's column block)"""
    n0, n1 = 96, 80
    ins = [np.zeros((fa.slab_local_size([n0, n1], 3, g)[1], n1), dtype=complex) for g in range(3)]
    sp = fa.SlabPlanC([n0, n1], [0, 0, 0], ins, ins, fa.FORWARD)
    assert fa.lib.fftw_amd_slab_num_devices(sp.handle) == 3
    assert "batch=32" in sp.local_plan_sprint(0, 0)                  # 32 rows of 80 points on every device
    assert sp.local_plan_sprint(2, 1) is not None                    # columns of the last block: 26 wide
    with pytest.raises(ValueError):
        fa.SlabPlanC([n0], [0], ins[:1], ins[:1], fa.FORWARD)         # rank 1 is not a slab problem
    with pytest.raises(ValueError):
        fa.SlabPlanC([n0, 0], [0], ins[:1], ins[:1], fa.FORWARD)
    if fa.device_count() == 0:
        with pytest.raises(RuntimeError):
            sp.execute()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,ndev", [((96, 80), 3), ((1024, 512), 2), ((64, 64), 1), ((7, 5), 4), ((40, 24, 16), 2),
                                        ((128, 128, 128), 2), ((4096, 4096), 2)])
def test_slab_transform_matches_the_single_device_plan(shape, ndev):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(sum(shape) + ndev)
    n = int(np.prod(shape))
    x = crand(rng, 1, n)
    rest = n // shape[0]
    cuts = [fa.slab_local_size(list(shape), ndev, g)[1:] for g in range(ndev)]
    for sign in (-1, 1):
        # reference: the whole transform on one device (itself checked against the oracle elsewhere); small shapes: the oracle
        xd = torch.from_numpy(x).to(dev)
        yd = torch.zeros_like(xd)
        fa.plan_many_dft(len(shape), list(shape), 1, xd, None, 1, n, yd, None, 1, n, sign).execute()
        torch.cuda.synchronize()
        want = yd.cpu().numpy().reshape(shape[0], rest)
        if n <= 1 << 16:
            assert aerror(want.reshape(1, n), oracle_dft(x, shape, 1, sign).reshape(1, n)) < TOL
        ins = [torch.from_numpy(x.reshape(shape[0], rest)[lo:lo + ln0].copy()).to(dev) if ln0 else torch.zeros(1, dtype=torch.complex128, device=dev)
               for ln0, lo in cuts]
        outs = [torch.zeros_like(t) for t in ins]
        sp = fa.SlabPlanC(list(shape), [0] * ndev, ins, outs, sign)
        sp.execute()
        sp.execute()                                                  # a second run right behind the first
        sp.sync()
        for (ln0, lo), o in zip(cuts, outs):
            if ln0:
                assert aerror(o.cpu().numpy().reshape(ln0, rest), want[lo:lo + ln0]) < TOL, (shape, ndev, sign)
        # in place
        sp2 = fa.SlabPlanC(list(shape), [0] * ndev, ins, ins, sign)
        sp2.execute()
        sp2.sync()
        for (ln0, lo), o in zip(cuts, ins):
            if ln0:
                assert aerror(o.cpu().numpy().reshape(ln0, rest), want[lo:lo + ln0]) < TOL, (shape, ndev, sign, "in place")
        sp.destroy()
        sp2.destroy()


C_SLAB = r"""
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <fftw3.h>
#include <fftw3_amd.h>
/* a C caller spreads one 192 x 160 transform over "three devices" (device 0 three times on this box) and checks it
   against the same transform run by ONE ordinary plan of the library */
int main(void) {
    const long long n[2] = {192, 160};
    const int ndev = 3, devs[3] = {0, 0, 0};
    const size_t total = (size_t)(n[0] * n[1]);
    fftw_complex *h = (fftw_complex *)malloc(total * sizeof(fftw_complex)), *ref = (fftw_complex *)malloc(total * sizeof(fftw_complex));
    fftw_complex *got = (fftw_complex *)malloc(total * sizeof(fftw_complex));
    fftw_complex *in[3], *out[3];
    long long ln0, lo;
    size_t i;
    int g;
    if (fftw_amd_device_count() < 1) { printf("no device\n"); return 2; }
    srand48(1);
    for (i = 0; i < total; ++i) { h[i][0] = drand48() - 0.5; h[i][1] = drand48() - 0.5; }
    fftw_plan p = fftw_plan_dft_2d((int)n[0], (int)n[1], h, ref, FFTW_FORWARD, FFTW_ESTIMATE);    /* host arrays: staged */
    fftw_execute(p);
    fftw_destroy_plan(p);
    for (g = 0; g < ndev; ++g) {
        long long elems = fftw_amd_slab_local_size(2, n, ndev, g, &ln0, &lo);
        in[g] = (fftw_complex *)fftw_amd_malloc_device((size_t)(elems ? elems : 1) * sizeof(fftw_complex));
        out[g] = (fftw_complex *)fftw_amd_malloc_device((size_t)(elems ? elems : 1) * sizeof(fftw_complex));
        fftw_amd_memcpy_to_device(in[g], h + lo * n[1], (size_t)elems * sizeof(fftw_complex));
    }
    fftw_amd_slab_plan sp = fftw_amd_slab_plan_dft(2, n, ndev, devs, in, out, FFTW_FORWARD, FFTW_ESTIMATE);
    if (!sp) { printf("slab planner returned NULL\n"); return 3; }
    fftw_amd_slab_execute(sp);
    fftw_amd_slab_sync(sp);
    for (g = 0; g < ndev; ++g) {
        long long elems = fftw_amd_slab_local_size(2, n, ndev, g, &ln0, &lo);
        fftw_amd_memcpy_to_host(got + lo * n[1], out[g], (size_t)elems * sizeof(fftw_complex));
    }
    double worst = 0.0, scale = 0.0;
    for (i = 0; i < total; ++i) {
        double dr = fabs(got[i][0] - ref[i][0]), di = fabs(got[i][1] - ref[i][1]);
        double a = fabs(ref[i][0]) > fabs(ref[i][1]) ? fabs(ref[i][0]) : fabs(ref[i][1]);
        if (dr > worst) worst = dr;
        if (di > worst) worst = di;
        if (a > scale) scale = a;
    }
    if (!(worst <= 1e-12 * scale)) { printf("slab result differs: %g (scale %g)\n", worst, scale); return 5; }
    fftw_amd_destroy_slab_plan(sp);
    for (g = 0; g < ndev; ++g) { fftw_amd_free_device(in[g]); fftw_amd_free_device(out[g]); }
    free(h); free(ref); free(got);
    printf("slab client ok\n");
    return 0;
}
"""


@pytest.mark.gpu
def test_c_client_spreads_one_transform_over_three_streams_of_one_device(tmp_path):
    import os
    import subprocess
    from util import ROOT
    src = tmp_path / "slabc.c"
    exe = tmp_path / "slabc"
    src.write_text(C_SLAB)
    libdir = os.path.join(ROOT, "fftw3_amd", "lib")
    subprocess.run(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir,
                    "-lfftw3", "-Wl,-rpath," + libdir, "-lm", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "slab client ok" in r.stdout, r.stdout
