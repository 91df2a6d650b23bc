"""gloo worker of tests/test_distributed.py: slab-decomposed transforms (fftw3_amd.slab) on
WORLD_SIZE CPU ranks; local plans run through the numpy step interpreter (no GPU here),
on a GPU node the same SlabPlan executes them on the rank's GPU and exchanges over RCCL."""
import os, sys
sys.path.insert(0, os.environ["FA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["FA_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
import fftw3_amd as fa
from fftw3_amd import slab
from step_interp import run_plan_on_host
from util import oracle_dft, oracle_r2c, oracle_c2r, oracle_r2r, aerror, TOL
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
DEV = os.environ.get("FA_SLAB_DEVICE", "cpu")     # "cuda": ranks share cuda:0, plans run on the GPU
def host_exec(plan, src, dst):
    run_plan_on_host(plan, src.numpy(), dst.numpy())
EXEC = None if DEV == "cuda" else host_exec
_zeros = torch.zeros
def zeros(n, dtype):
    return _zeros(n, dtype=dtype, device=DEV)
def put(t, arr):
    t[:arr.size] = torch.from_numpy(arr).to(DEV)
def get(t):
    if DEV == "cuda":
        torch.cuda.synchronize()
    return t.cpu().numpy()
def take(full, axis_t, s, cnt):
    a = np.swapaxes(full, 0, 1) if axis_t else full
    return np.ascontiguousarray(a[s:s + cnt]).reshape(-1)
def check_c2c(n, hm, flags, sign=fa.FORWARD):
    rng = np.random.default_rng(5)
    tot = int(np.prod(n)) * hm
    full = ((rng.random(tot) - 0.5) + 1j * (rng.random(tot) - 0.5)).reshape(list(n) + [hm])
    alloc, ln0, s0, ln1, s1 = slab.local_size_many_transposed(n, hm, 0, 0, world, rank)
    x = zeros(max(1, alloc), torch.complex128); y = zeros(max(1, alloc), torch.complex128)
    tin, tout = bool(flags & slab.TRANSPOSED_IN), bool(flags & slab.TRANSPOSED_OUT)
    loc = take(full, tin, s1 if tin else s0, ln1 if tin else ln0)
    put(x, loc)
    p = slab.plan_many_dft(n, hm, 0, 0, x, y, sign, fa.ESTIMATE | flags, executor=EXEC)
    p.execute()
    want = np.empty_like(full)
    for h in range(hm):
        want[..., h] = oracle_dft(np.ascontiguousarray(full[..., h]).reshape(-1), tuple(n), 1, sign=sign).reshape(n)
    wl = take(want, tout, s1 if tout else s0, ln1 if tout else ln0)
    e = aerror(get(y)[:wl.size], wl) if wl.size else 0.0
    assert e < TOL, ("c2c", n, hm, flags, e)
def check_r2r(n, hm, flags, kinds):
    rng = np.random.default_rng(6)
    tot = int(np.prod(n)) * hm
    full = (rng.random(tot) - 0.5).reshape(list(n) + [hm])
    alloc, ln0, s0, ln1, s1 = slab.local_size_many_transposed(n, hm, 0, 0, world, rank)
    x = zeros(max(1, alloc), torch.float64); y = zeros(max(1, alloc), torch.float64)
    tin, tout = bool(flags & slab.TRANSPOSED_IN), bool(flags & slab.TRANSPOSED_OUT)
    loc = take(full, tin, s1 if tin else s0, ln1 if tin else ln0)
    put(x, loc)
    p = slab.plan_many_r2r(n, hm, 0, 0, x, y, kinds, fa.ESTIMATE | flags, executor=EXEC)
    p.execute()
    want = np.empty_like(full)
    for h in range(hm):
        want[..., h] = oracle_r2r(np.ascontiguousarray(full[..., h]).reshape(-1), list(n), kinds).reshape(n)
    wl = take(want, tout, s1 if tout else s0, ln1 if tout else ln0)
    e = aerror(get(y)[:wl.size], wl) if wl.size else 0.0
    assert e < TOL, ("r2r", n, hm, flags, e)
def check_real(n, hm, tflag):
    """r2c (optionally TRANSPOSED_OUT) then c2r (optionally TRANSPOSED_IN) round trip + oracle"""
    rng = np.random.default_rng(7)
    tot = int(np.prod(n)) * hm
    full = (rng.random(tot) - 0.5).reshape(list(n) + [hm])
    nh = n[-1] // 2 + 1
    ne = list(n[:-1]) + [nh]
    alloc, ln0, s0, ln1, s1 = slab.local_size_many_transposed(ne, hm, 0, 0, world, rank)
    xr = zeros(max(2, 2 * alloc), torch.float64)
    yc = zeros(max(1, alloc), torch.complex128)
    pad = np.zeros([ln0] + list(n[1:-1]) + [2 * nh, hm])
    pad[..., :n[-1], :] = full[s0:s0 + ln0]
    put(xr, pad.reshape(-1))
    p = slab.plan_many_dft_r2c(n, hm, 0, 0, xr, yc, fa.ESTIMATE | (slab.TRANSPOSED_OUT if tflag else 0), executor=EXEC)
    p.execute()
    want = np.empty(ne + [hm], dtype=np.complex128)
    for h in range(hm):
        want[..., h] = oracle_r2c(np.ascontiguousarray(full[..., h]).reshape(-1), tuple(n), 1).reshape(ne)
    wl = take(want, tflag, s1 if tflag else s0, ln1 if tflag else ln0)
    e = aerror(get(yc)[:wl.size], wl) if wl.size else 0.0
    assert e < TOL, ("r2c", n, hm, tflag, e)
    # back
    zr = zeros(max(2, 2 * alloc), torch.float64)
    q = slab.plan_many_dft_c2r(n, hm, 0, 0, yc, zr, fa.ESTIMATE | (slab.TRANSPOSED_IN if tflag else 0), executor=EXEC)
    q.execute()
    got = get(zr)[:pad.size].reshape(pad.shape)[..., :n[-1], :]
    ref = full[s0:s0 + ln0] * np.prod(n)
    e = aerror(got, ref) if ref.size else 0.0
    assert e < TOL, ("c2r", n, hm, tflag, e)
for n, hm in [([8, 6], 1), ([7, 9], 2), ([5, 4, 6], 1), ([16, 3, 2, 2], 3), ([3, 32], 1), ([1, 5], 1)]:
    for flags in (0, slab.TRANSPOSED_OUT, slab.TRANSPOSED_IN, slab.TRANSPOSED_IN | slab.TRANSPOSED_OUT):
        check_c2c(n, hm, flags)
check_c2c([12, 10], 1, 0, fa.BACKWARD)
for n, hm, kinds in [([8, 6], 1, [5, 3]), ([7, 9], 2, [10, 0]), ([5, 4, 6], 1, [4, 8, 2])]:
    for flags in (0, slab.TRANSPOSED_OUT, slab.TRANSPOSED_IN):
        check_r2r(n, hm, flags, kinds)
for n, hm in [([8, 6], 1), ([7, 9], 2), ([5, 4, 6], 1), ([6, 3, 5], 2), ([4, 16], 1)]:
    for t in (False, True):
        check_real(n, hm, t)
dist.barrier(); dist.destroy_process_group()
print("rank %d ok" % rank)
