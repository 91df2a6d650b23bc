"""Batch sharding of one large `howmany` across the GPUs of a node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The
transforms of a batch are independent, so rank g owns a contiguous block of
whole batch elements and runs an ordinary single-GPU plan on it -- there is no
exchange inside a transform.  The only collective is the optional all-gather
that reassembles the outputs on every rank (north_star); it is xGMI-bound and
is therefore a separate, explicitly requested step.

Block partition: rank g gets [g*ceil(B/G), min(B, (g+1)*ceil(B/G))), the same
rule the reference uses to split a vector loop over threads
(fftw/threads/dft-vrank-geq1.c:158-159) and over MPI ranks (fftw/mpi/block.c:35-42).
"""
import fftw3_amd as fa


def shard_range(batch, world, rank):
    """[lo, hi) of the batch elements owned by `rank` of `world`"""
    block = (batch + world - 1) // world
    lo = min(batch, rank * block)
    hi = min(batch, lo + block)
    return lo, hi


class ShardedManyDft(object):
    """fftw_plan_many_dft over a batch that is split across ranks.

    `x_local` / `y_local` hold only this rank's shard (shape [hi-lo, *n]),
    contiguous.  execute() transforms the shard; all_gather() returns the full
    [batch, *n] output on every rank."""

    def __init__(self, n, batch, x_local, y_local, sign=fa.FORWARD, flags=fa.ESTIMATE,
                 world=1, rank=0):
        self.n = list(n)
        self.batch, self.world, self.rank = batch, world, rank
        self.lo, self.hi = shard_range(batch, world, rank)
        self.count = self.hi - self.lo
        size = 1
        for v in self.n:
            size *= v
        self.size = size
        self.y_local = y_local
        self.plan = fa.plan_many_dft(len(self.n), self.n, self.count, x_local, None, 1, size,
                                     y_local, None, 1, size, sign, flags)

    def execute(self):
        self.plan.execute()

    def all_gather(self, dist, out=None):
        """RCCL all-gather of the shards into [world*block, *n]; the tail beyond
        `batch` (uneven split) is padding."""
        import torch
        block = (self.batch + self.world - 1) // self.world
        y = self.y_local
        if self.count < block:      # uneven last shard: pad to the common block size
            pad = torch.zeros((block,) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
            pad[: self.count] = y
            y = pad
        if out is None:
            out = torch.empty((self.world * block,) + tuple(y.shape[1:]), dtype=y.dtype,
                              device=y.device)
        # complex dtypes travel as pairs of doubles (ncclDouble, count = 2*N*B/G per rank)
        src = torch.view_as_real(y) if y.is_complex() else y
        dst = torch.view_as_real(out) if out.is_complex() else out
        dist.all_gather_into_tensor(dst, src.contiguous())
        return out[: self.batch]
