"""Batch sharding inside the C ABI (include/fftw3_amd.h, fftw3_amd/csrc/sharded.c): one batched
transform cut over several devices with the reference's block rule
(fftw/threads/dft-vrank-geq1.c:158-159, fftw/mpi/block.c:35-42), one plan replica / stream / host
thread per device.  The one-GPU box runs it with the same device named twice (devs = {0, 0}): two
replicas, two streams, two host threads, peer-to-peer gather on the same card."""
import os
import subprocess

import numpy as np
import pytest

import fftw3_amd as fa
from util import ROOT, TOL, aerror, crand, oracle_dft, oracle_r2c, rrand


def test_block_rule_of_the_c_layer_matches_the_python_helper():
    from fftw3_amd.parallel import shard_range as py_rule
    for b in (0, 1, 7, 8, 512, 4096, 4097):
        for p in (1, 2, 3, 4, 8):
            cover = []
            for g in range(p):
                lo, hi = fa.shard_range(b, p, g)
                assert (lo, hi) == tuple(py_rule(b, p, g))
                cover += list(range(lo, hi))
            assert cover == list(range(b))


def test_sharded_plan_is_built_without_a_device_and_uses_one_replica_per_shard():
    n, b = 4096, 7
    ins = [np.zeros((4, n), dtype=complex), np.zeros((3, n), dtype=complex)]
    outs = [np.zeros_like(a) for a in ins]
    sp = fa.plan_many_dft_sharded(1, [n], b, [0, 0], ins, None, 1, n, outs, None, 1, n, fa.FORWARD)
    assert sp.num_shards == 2 and [sp.device(g) for g in range(2)] == [0, 0]
    assert [sp.range(g) for g in range(2)] == [(0, 4), (4, 7)]
    assert "batch=4" in sp.replica_sprint(0) and "batch=3" in sp.replica_sprint(1)
    # more shards than transforms: trailing shards are empty and own no replica (mpi/block.c:39-50)
    sp3 = fa.plan_many_dft_sharded(1, [n], 2, [0, 0, 0], ins + [None], None, 1, n, outs + [None], None, 1, n, fa.FORWARD)
    assert [sp3.range(g) for g in range(3)] == [(0, 1), (1, 2), (2, 2)] and sp3.replica_sprint(2) is None
    if fa.device_count() == 0:
        with pytest.raises(RuntimeError):
            sp.execute()


def test_gather_schedule_all_gather_for_equal_shards_broadcasts_for_ragged_tails():
    """argument layout of the RCCL gather (fftw_amd_sharded_gather_ops: exactly the calls
    fftw_amd_sharded_all_gather issues inside its group), checked without a device: equal shards -> one
    ncclAllGather per rank with the shard's byte count; a ragged block-rule tail -> one ncclBroadcast per
    non-empty shard and rank, root = the shard, receive pointer = that shard's place in the rank's image."""
    n = 256
    odist_bytes = n * 16

    def plan(b, ndev):
        cuts = [fa.shard_range(b, ndev, g) for g in range(ndev)]
        ins = [np.zeros((max(1, hi - lo), n), dtype=complex) if hi > lo else None for lo, hi in cuts]
        outs = [np.zeros_like(a) if a is not None else None for a in ins]
        sp = fa.plan_many_dft_sharded(1, [n], b, list(range(ndev)), ins, None, 1, n, outs, None, 1, n, fa.FORWARD)
        return sp, cuts, outs

    # 8 transforms on 4 devices: equal shards of 2
    sp, cuts, outs = plan(8, 4)
    full = [0x10000000 * (d + 1) for d in range(4)]
    ops = sp.gather_ops(full)
    assert len(ops) == 4
    for d, (kind, rank, root, send, recv, nbytes) in enumerate(ops):
        assert (kind, rank, root) == (0, d, -1)
        assert send == outs[d].ctypes.data and recv == full[d] and nbytes == 2 * odist_bytes
    # 7 transforms on 4 devices: shards 2, 2, 2, 1 -> broadcasts
    sp, cuts, outs = plan(7, 4)
    ops = sp.gather_ops(full)
    assert len(ops) == 16
    k = 0
    for g, (lo, hi) in enumerate(cuts):
        for d in range(4):
            kind, rank, root, send, recv, nbytes = ops[k]
            k += 1
            assert (kind, rank, root) == (1, d, g)
            assert send == outs[g].ctypes.data and recv == full[d] + lo * odist_bytes and nbytes == (hi - lo) * odist_bytes
    # 2 transforms on 3 devices: the empty trailing shard takes part in no call
    sp, cuts, outs = plan(2, 3)
    ops = sp.gather_ops(full[:3])
    assert len(ops) == 6 and {o[2] for o in ops} == {0, 1} and all(o[5] == odist_bytes for o in ops)


def test_rccl_entry_points_resolve():
    """the loader of the gather (sharded.c rccl_load_once) opens librccl.so once per process and resolves the
    seven entry points it calls; no RCCL call is made (works without a GPU)"""
    if not any(os.path.exists(os.path.join(d, "librccl.so")) for d in ("/opt/rocm/lib", "/usr/lib", "/usr/local/lib")):
        pytest.skip("no librccl.so on this machine")
    assert fa.lib.fftw_amd_rccl_probe() == 7
    assert fa.lib.fftw_amd_rccl_probe() == 7      # second call: same table, no second dlopen


@pytest.mark.gpu
def test_replicas_live_on_their_shards_device():
    """every replica's tables and scratch are allocated on devs[g] (the planner runs with that device current:
    mk_sharded), not on whatever device the calling thread had selected; the caller's device is restored.
    With >= 2 visible devices the shards go to devices 0 and 1, otherwise both to device 0."""
    import torch
    ndevs = fa.device_count()
    devs = [0, 1] if ndevs >= 2 else [0, 0]
    n, b = 1 << 16, 6
    ins = [torch.zeros(3, n, dtype=torch.complex128, device="cuda:%d" % d) for d in devs]
    outs = [torch.zeros_like(t) for t in ins]
    if ndevs >= 2:
        fa.lib.fftw_amd_set_device(ndevs - 1)      # a current device that is neither shard's (or the last one)
    before = fa.lib.fftw_amd_get_device()
    sp = fa.plan_many_dft_sharded(1, [n], b, devs, ins, None, 1, n, outs, None, 1, n, fa.FORWARD)
    assert fa.lib.fftw_amd_get_device() == before
    assert [sp.replica_device(g) for g in range(2)] == devs
    sp.execute()
    sp.sync()
    fa.lib.fftw_amd_set_device(0)
    # a device that does not exist is refused at plan time
    with pytest.raises(ValueError):
        fa.plan_many_dft_sharded(1, [n], b, [0, ndevs + 3], ins, None, 1, n, outs, None, 1, n, fa.FORWARD)


@pytest.mark.gpu
def test_two_shards_on_one_device_match_the_oracle_and_gather():
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    n, b = 1 << 16, 7
    x = crand(rng, b, n)
    want = oracle_dft(x, (n,), b).reshape(b, n)
    cuts = [fa.shard_range(b, 2, g) for g in range(2)]
    ins = [torch.from_numpy(x[lo:hi].copy()).to(dev) for lo, hi in cuts]
    outs = [torch.zeros_like(t) for t in ins]
    sp = fa.plan_many_dft_sharded(1, [n], b, [0, 0], ins, None, 1, n, outs, None, 1, n, fa.FORWARD)
    sp.execute()
    sp.sync()
    for (lo, hi), o in zip(cuts, outs):
        assert aerror(o.cpu().numpy(), want[lo:hi]) < TOL
    # reassemble on every "device": peer-to-peer pushes (RCCL needs distinct devices)
    full = [torch.zeros((b, n), dtype=torch.complex128, device=dev) for _ in range(2)]
    assert sp.all_gather(full, 0) == 0
    sp.sync()
    for f in full:
        assert aerror(f.cpu().numpy(), want) < TOL
    # execute again (replicas, streams and threads are reused), r2c through the same layer
    sp.execute()
    sp.sync()
    assert aerror(outs[1].cpu().numpy(), want[cuts[1][0]:cuts[1][1]]) < TOL
    sp.destroy()
    nr, br = 8192, 5
    xr = rrand(rng, br, nr)
    wr = oracle_r2c(xr, (nr,), br).reshape(br, nr // 2 + 1)
    cuts = [fa.shard_range(br, 3, g) for g in range(3)]
    ins = [torch.from_numpy(xr[lo:hi].copy()).to(dev) for lo, hi in cuts]
    outs = [torch.zeros((hi - lo, nr // 2 + 1), dtype=torch.complex128, device=dev) for lo, hi in cuts]
    sr = fa.plan_many_dft_r2c_sharded(1, [nr], br, [0, 0, 0], ins, None, 1, nr, outs, None, 1, nr // 2 + 1)
    sr.execute()
    sr.sync()
    for (lo, hi), o in zip(cuts, outs):
        assert aerror(o.cpu().numpy(), wr[lo:hi]) < TOL


@pytest.mark.gpu
def test_single_shard_gather_goes_through_rccl():
    """ndev = 1: the RCCL path (dlopen, ncclCommInitAll, grouped ncclBroadcast) with one rank"""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(12)
    n, b = 4096, 6
    x = crand(rng, b, n)
    xi = torch.from_numpy(x).to(dev)
    yo = torch.zeros_like(xi)
    sp = fa.plan_many_dft_sharded(1, [n], b, [0], [xi], None, 1, n, [yo], None, 1, n, fa.FORWARD)
    sp.execute()
    full = [torch.zeros_like(xi)]
    rc = sp.all_gather(full, 0)
    sp.sync()
    torch.cuda.synchronize()
    assert rc in (0, 1)
    assert aerror(full[0].cpu().numpy(), oracle_dft(x, (n,), b).reshape(b, n)) < TOL
    if rc != 1:
        pytest.skip("librccl.so could not be initialised on this box: peer-to-peer path verified instead")


C_SHARDED = r"""
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fftw3.h>
#include <fftw3_amd.h>
/* a C caller shards one fftw_plan_many_dft batch over "two devices" (device 0 twice on this box)
   and checks it against the same batch run by ONE ordinary plan of the library */
int main(void) {
    const int n = 1 << 14, howmany = 9, ndev = 2;
    int devs[2] = {0, 0}, nn[1] = {n};
    size_t bytes = (size_t)howmany * n * sizeof(fftw_complex);
    fftw_complex *h = (fftw_complex *)malloc(bytes), *r1 = (fftw_complex *)malloc(bytes), *r2 = (fftw_complex *)malloc(bytes);
    fftw_complex *in[2], *out[2], *full[2];
    long long lo, hi;
    int g, i;
    if (fftw_amd_device_count() < 1) { printf("no device\n"); return 2; }
    srand48(1);
    for (i = 0; i < howmany * n; ++i) { h[i][0] = drand48() - 0.5; h[i][1] = drand48() - 0.5; }
    /* reference run: one plan, host arrays (staged) */
    fftw_plan p = fftw_plan_many_dft(1, nn, howmany, h, NULL, 1, n, r1, NULL, 1, n, FFTW_FORWARD, FFTW_ESTIMATE);
    fftw_execute(p);
    fftw_destroy_plan(p);
    /* sharded run on device memory */
    for (g = 0; g < ndev; ++g) {
        fftw_amd_shard_range(howmany, ndev, g, &lo, &hi);
        size_t sb = (size_t)(hi - lo) * n * sizeof(fftw_complex);
        in[g] = (fftw_complex *)fftw_amd_malloc_device(sb ? sb : 16);
        out[g] = (fftw_complex *)fftw_amd_malloc_device(sb ? sb : 16);
        full[g] = (fftw_complex *)fftw_amd_malloc_device(bytes);
    }
    fftw_amd_sharded_plan sp = fftw_amd_plan_many_dft_sharded(1, nn, howmany, ndev, devs, in, NULL, 1, n, out, NULL, 1, n,
                                                             FFTW_FORWARD, FFTW_ESTIMATE);
    if (!sp) { printf("sharded planner returned NULL\n"); return 3; }
    for (g = 0; g < ndev; ++g) {
        fftw_amd_sharded_range(sp, g, &lo, &hi);
        fftw_amd_memcpy_to_device(in[g], h + lo * n, (size_t)(hi - lo) * n * sizeof(fftw_complex));
    }
    fftw_amd_execute_sharded(sp);
    if (fftw_amd_sharded_all_gather(sp, (void *const *)full, 0) < 0) { printf("gather failed\n"); return 4; }
    fftw_amd_sharded_sync(sp);
    for (g = 0; g < ndev; ++g) {
        double worst = 0.0, scale = 0.0;
        fftw_amd_memcpy_to_host(r2, full[g], bytes);
        for (i = 0; i < howmany * n; ++i) {
            double dr = r2[i][0] - r1[i][0], di = r2[i][1] - r1[i][1];
            double a = fabs(r1[i][0]) > fabs(r1[i][1]) ? fabs(r1[i][0]) : fabs(r1[i][1]);
            if (fabs(dr) > worst) worst = fabs(dr);
            if (fabs(di) > worst) worst = fabs(di);
            if (a > scale) scale = a;
        }
        if (!(worst <= 1e-12 * scale)) { printf("image %d differs: %g (scale %g)\n", g, worst, scale); return 5; }
    }
    fftw_amd_destroy_sharded_plan(sp);
    for (g = 0; g < ndev; ++g) { fftw_amd_free_device(in[g]); fftw_amd_free_device(out[g]); fftw_amd_free_device(full[g]); }
    free(h); free(r1); free(r2);
    printf("sharded client ok\n");
    return 0;
}
"""


@pytest.mark.gpu
def test_c_client_shards_a_batch_over_two_streams_of_one_device(tmp_path):
    src = tmp_path / "sharded.c"
    exe = tmp_path / "sharded"
    src.write_text(C_SHARDED)
    libdir = os.path.join(ROOT, "fftw3_amd", "lib")
    subprocess.run(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir,
                    "-lfftw3", "-Wl,-rpath," + libdir, "-lm", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "sharded client ok" in r.stdout, r.stdout
