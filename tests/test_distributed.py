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
