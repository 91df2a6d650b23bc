"""N > 1 path on CPU: two gloo ranks shard a batch (fftw3_amd.parallel), each
transforms its shard, the outputs are all-gathered and compared with the
oracle on the full batch.  The ranks have no GPU here, so each interprets its
plan's step list with tests/step_interp.py; on a GPU node the same code runs
plan.execute() and the same collective over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np

from fftw3_amd.parallel import shard_range
from util import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["FA_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
import fftw3_amd as fa
from fftw3_amd.parallel import ShardedManyDft, shard_range
from step_interp import run_plan_on_host
from util import oracle_dft, aerror, TOL
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, batch = 4096, 7                      # uneven split: 4 + 3
rng = np.random.default_rng(123)
full = (rng.random((batch, n)) - 0.5) + 1j * (rng.random((batch, n)) - 0.5)   # same on every rank
lo, hi = shard_range(batch, world, rank)
x = torch.from_numpy(full[lo:hi].copy())
y = torch.zeros_like(x)
sh = ShardedManyDft([n], batch, x, y, fa.FORWARD, fa.ESTIMATE, world, rank)
assert (sh.lo, sh.hi) == (lo, hi) and sh.plan.batch == hi - lo
run_plan_on_host(sh.plan, x.numpy(), y.numpy())         # GPU ranks call sh.execute() here
out = sh.all_gather(dist)
assert out.shape == (batch, n)
e = aerror(out.numpy(), oracle_dft(full, (n,), batch).reshape(batch, n))
assert e < TOL, e
# max-over-ranks reduction used by bench.py for the timing
t = torch.tensor([1.0 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == float(world)
dist.barrier()
dist.destroy_process_group()
print("rank %d ok %.2e" % (rank, e))
'''


def test_shard_range_is_a_partition():
    for batch in (0, 1, 7, 8, 512, 4096, 4097):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(batch, world, r)
                assert 0 <= lo <= hi <= batch
                cover += list(range(lo, hi))
            assert cover == list(range(batch))
            sizes = [shard_range(batch, world, r)[1] - shard_range(batch, world, r)[0] for r in range(world)]
            assert max(sizes) == (batch + world - 1) // world          # block rule of mpi/block.c:35-42


def _run_ranks(script, world, timeout=300):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(world):
        env = dict(os.environ, FA_ROOT=ROOT, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
        assert "rank %d ok" % rank in out


def test_slab_local_size_matches_reference_block_rule():
    """fftw_mpi_local_size_*: blocks of ceil(n/P), trailing ranks may be empty (mpi/block.c:39-50)"""
    from fftw3_amd import slab
    for n0, n1, world in [(8, 6, 2), (7, 9, 3), (5, 4, 8), (4096, 4096, 8), (3, 32, 4)]:
        rows = cols = 0
        for r in range(world):
            alloc, ln0, s0, ln1, s1 = slab.local_size_2d_transposed(n0, n1, world, r)
            assert s0 == min(n0, r * ((n0 + world - 1) // world)) and s1 == min(n1, r * ((n1 + world - 1) // world))
            assert alloc == max(ln0 * n1, ln1 * n0)
            rows += ln0
            cols += ln1
        assert rows == n0 and cols == n1
    a, ln0, s0 = slab.local_size_3d(10, 6, 4, 4, 3)
    assert (ln0, s0) == (1, 9) and a == max(1 * 6, 0 * 10) * 4


def test_slab_world1_without_process_group():
    """world = 1: the exchange degenerates to a copy; every kind against the oracle"""
    import torch
    from fftw3_amd import slab
    from step_interp import run_plan_on_host
    from util import oracle_dft, oracle_r2r, aerror, TOL

    def host_exec(plan, src, dst):
        run_plan_on_host(plan, src.numpy(), dst.numpy())
    rng = np.random.default_rng(2)
    n = [6, 10, 4]
    full = (rng.random(n) - 0.5) + 1j * (rng.random(n) - 0.5)
    x = torch.from_numpy(full.reshape(-1).copy())
    y = torch.zeros_like(x)
    p = slab.plan_dft_3d(6, 10, 4, x, y, fa_sign_forward(), executor=host_exec, world=1, rank=0)
    p.execute()
    assert p.num_exchanges() == 2
    assert aerror(y.numpy(), oracle_dft(full.reshape(-1), tuple(n), 1)) < TOL
    xr = torch.from_numpy((rng.random(60) - 0.5))
    yr = torch.zeros_like(xr)
    q = slab.plan_r2r_2d(6, 10, xr, yr, 5, 7, slab.TRANSPOSED_OUT | 64, executor=host_exec, world=1, rank=0)
    q.execute()
    assert q.num_exchanges() == 1
    want = oracle_r2r(xr.numpy(), [6, 10], [5, 7]).reshape(6, 10).T
    assert aerror(yr.numpy(), want.reshape(-1)) < TOL


def fa_sign_forward():
    import fftw3_amd as fa
    return fa.FORWARD


def test_two_rank_gloo_slab_transforms():
    _run_ranks(os.path.join(ROOT, "tests", "workers", "slab_worker.py"), 2)


def test_three_rank_gloo_slab_transforms_uneven_blocks():
    _run_ranks(os.path.join(ROOT, "tests", "workers", "slab_worker.py"), 3)


def test_two_rank_gloo_sharded_transform_and_gather(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, FA_ROOT=ROOT, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
        assert "rank %d ok" % rank in out


LAUNCH_WORKER = r'''
import json, os, sys
import torch, torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert sys.argv[1:] == ["--gpus", str(world), "--steps", "2"], sys.argv
t = torch.tensor([float(rank)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
if os.environ.get("FAIL_RANK") == str(rank):
    sys.exit(3)
if rank == 0:
    print(json.dumps({"n_gpus": world, "max_rank": t.item(), "local_rank": os.environ["LOCAL_RANK"]}))
else:
    print("noise from rank", rank)          # must not reach the launcher's stdout
dist.destroy_process_group()
'''


def test_bench_launcher_spawns_ranks_and_relays_rank0(tmp_path, capsys, monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE must start its own N ranks
    (the driver's command form), relay rank 0's single line and fail if a rank fails."""
    import json
    import bench
    monkeypatch.delenv("MASTER_PORT", raising=False)
    script = tmp_path / "w.py"
    script.write_text(LAUNCH_WORKER)
    rc = bench.spawn_ranks(2, ["--gpus", "2", "--steps", "2"], script=str(script), timeout=240)
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1, out
    line = json.loads(out[0])
    assert line == {"n_gpus": 2, "max_rank": 1.0, "local_rank": "0"}
    monkeypatch.setenv("FAIL_RANK", "1")
    rc = bench.spawn_ranks(2, ["--gpus", "2", "--steps", "2"], script=str(script), timeout=240)
    assert rc != 0


def test_bench_main_takes_the_launcher_branch_before_touching_torch(monkeypatch):
    """--gpus 2 and no WORLD_SIZE: main() must hand over to spawn_ranks before importing
    torch.cuda / fftw3_amd (a process that touched the GPU must not fork ranks)."""
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen = {}

    def fake(n, argv, **kw):
        seen["n"], seen["argv"] = n, list(argv)
        return 0
    monkeypatch.setattr(bench, "spawn_ranks", fake)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen == {"n": 2, "argv": ["--gpus", "2", "--steps", "1", "--warmup", "0"]}
