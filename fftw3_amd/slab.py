"""Slab-decomposed multi-dimensional transforms across the GPUs of a node
(SURVEY.md section 8(f) row 4): one transform too large for, or wanted across,
several GPUs.  Host-side mirror of the reference's distributed-memory layer
(fftw/mpi/fftw3-mpi.h:74-215) with torch.distributed in the place of MPI: one
process per GPU, backend "nccl" = RCCL over xGMI (or "gloo" on CPU in tests).

Data distribution is the reference's (fftw/mpi/block.c:35-56,
api.c:354-406): the first dimension is split into blocks of
default_block(n0) = ceil(n0 / world) rows, rank g owning rows
[g*block, min(n0, (g+1)*block)); with TRANSPOSED_OUT / TRANSPOSED_IN the array
is instead split along the second dimension and stored [local_n1][n0][rest].
`howmany` transforms are interleaved (the innermost index), as in
fftw_mpi_plan_many_dft.  Real-data transforms keep the reference's padded real
layout: last dimension 2*(n_last/2+1) doubles (fftw/mpi/rdft2-rank-geq2.c).

Algorithm (reference fftw/mpi/dft-rank-geq2.c:60-120, transpose-alltoall.c:60-140,
re-laid out for one all-to-all per redistribution and no separate pack pass):

    1. local plan over every locally complete dimension, written through the
       guru strides directly in "destination-major" order, so that the part
       meant for rank r is one contiguous run of the send buffer;
    2. one all_to_all_single (RCCL: direct peer-to-peer over the xGMI mesh,
       no ring);
    3. strided copies (rank-0 guru plans = the library's copy kernel) that
       interleave the received blocks into the new slab;
    4. local plan along the dimension that has just become local;
    5. unless TRANSPOSED_OUT: steps 1-3 again in the other direction.

Every local transform and copy is an ordinary fftw3_amd plan executed through
the C-ABI on the rank's GPU; only the exchange is torch.distributed.
"""
import fftw3_amd as fa

DEFAULT_BLOCK = 0
TRANSPOSED_IN = 1 << 29      # fftw/mpi/fftw3-mpi.h:214
TRANSPOSED_OUT = 1 << 30     # fftw/mpi/fftw3-mpi.h:215
_MPI_FLAGS = TRANSPOSED_IN | TRANSPOSED_OUT


def default_block(n, world):
    """fftw_mpi_default_block (fftw/mpi/block.c:39-42)"""
    return (n + world - 1) // world


def block(n, blk, which):
    """extent of block `which` of a dimension of n cut into blocks of blk (fftw/mpi/block.c:46-50)"""
    d = n - which * blk
    return 0 if d <= 0 else min(blk, d)


def local_size_many_transposed(n, howmany, block0, block1, world, rank):
    """fftw_mpi_local_size_many_transposed: (alloc_local, local_n0, local_0_start,
    local_n1, local_1_start); alloc_local counts elements of the distributed
    array (complex numbers for dft; for r2c / c2r pass the complex extents, i.e.
    n_last/2+1 as the last entry, and allocate 2*alloc_local doubles for the
    real side, as the reference's manual prescribes)."""
    n = list(n)
    if len(n) == 0:
        return howmany, 1, 0, 1, 0
    b0 = block0 or default_block(n[0], world)
    if b0 * world < n[0]:
        raise ValueError("block0 too small for %d ranks" % world)
    ln0, s0 = block(n[0], b0, rank), min(n[0], rank * b0)
    rest = howmany
    for v in n[2:]:
        rest *= v
    if len(n) == 1:
        return ln0 * howmany, ln0, s0, ln0, s0
    b1 = block1 or default_block(n[1], world)
    if b1 * world < n[1]:
        raise ValueError("block1 too small for %d ranks" % world)
    ln1, s1 = block(n[1], b1, rank), min(n[1], rank * b1)
    alloc = max(ln0 * n[1], ln1 * n[0]) * rest
    return alloc, ln0, s0, ln1, s1


def local_size_many(n, howmany, block0, world, rank):
    a, ln0, s0, _, _ = local_size_many_transposed(n, howmany, block0, DEFAULT_BLOCK, world, rank)
    return a, ln0, s0


def local_size_transposed(n, world, rank):
    return local_size_many_transposed(n, 1, DEFAULT_BLOCK, DEFAULT_BLOCK, world, rank)


def local_size(n, world, rank):
    return local_size_many(n, 1, DEFAULT_BLOCK, world, rank)


def local_size_2d(n0, n1, world, rank):
    return local_size([n0, n1], world, rank)


def local_size_3d(n0, n1, n2, world, rank):
    return local_size([n0, n1, n2], world, rank)


def local_size_2d_transposed(n0, n1, world, rank):
    return local_size_transposed([n0, n1], world, rank)


def local_size_3d_transposed(n0, n1, n2, world, rank):
    return local_size_transposed([n0, n1, n2], world, rank)


def _gpu_executor(plan, src, dst):
    plan.execute()


def _prod(v):
    p = 1
    for e in v:
        p *= e
    return p


class _Exchange(object):
    """one all-to-all: run r of the send buffer goes to rank r, the runs received
    are stored back to back in source order"""

    def __init__(self, send, recv, send_counts, recv_counts, group, world):
        self.send, self.recv = send, recv
        self.send_counts, self.recv_counts = send_counts, recv_counts
        self.group, self.world = group, world

    def run(self):
        import torch
        n_s, n_r = sum(self.send_counts), sum(self.recv_counts)
        if self.world == 1:
            self.recv[:n_r].copy_(self.send[:n_s])
            return
        import torch.distributed as dist
        s, r, mul = self.send[:n_s], self.recv[:n_r], 1
        if s.is_complex():                      # complex numbers travel as pairs of doubles
            s, r, mul = torch.view_as_real(s).reshape(-1), torch.view_as_real(r).reshape(-1), 2
        dist.all_to_all_single(r, s, [mul * c for c in self.recv_counts],
                               [mul * c for c in self.send_counts], group=self.group)


class SlabPlan(object):
    """Distributed plan: a list of local fftw3_amd plans and all-to-all exchanges.

    kind: "c2c" (fftw_mpi_plan_many_dft), "r2c", "c2r" (fftw_mpi_plan_many_dft_r2c /
    _c2r) or "r2r" (fftw_mpi_plan_many_r2r, `r2r_kinds` = one fftw_r2r_kind per dim).
    x_local / y_local: contiguous tensors of the local slab:
      c2c   complex128, alloc_local elements each
      r2c   x float64 with 2*alloc_local doubles (last dim padded to 2*(n_last/2+1)),
            y complex128 alloc_local; TRANSPOSED_OUT allowed
      c2r   the reverse; TRANSPOSED_IN allowed
      r2r   float64, alloc_local each
    Layouts: normal [local_n0][n1][n2..][howmany]; transposed [local_n1][n0][n2..][howmany]
    (for r2c / c2r the complex side uses n_last/2+1 as its last extent).
    executor(plan, src, dst) runs one local plan; the default executes on the GPU."""

    def __init__(self, kind, n, howmany, block0, block1, x_local, y_local, sign=fa.FORWARD,
                 flags=fa.ESTIMATE, r2r_kinds=None, group=None, world=None, rank=None,
                 executor=None):
        import torch
        import torch.distributed as dist
        n = [int(v) for v in n]
        if len(n) < 2:
            raise NotImplementedError("slab decomposition needs rank >= 2 "
                                      "(the reference's 1d distributed plan is a separate solver)")
        if kind not in ("c2c", "r2c", "c2r", "r2r"):
            raise ValueError(kind)
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.kind, self.n, self.howmany = kind, n, howmany
        self.world, self.rank, self.group = world, rank, group
        self.sign = sign
        self.mpi_flags = flags & _MPI_FLAGS
        self.flags = flags & ~_MPI_FLAGS
        self.r2r_kinds = list(r2r_kinds) if r2r_kinds is not None else None
        self.executor = executor or _gpu_executor
        tin = bool(self.mpi_flags & TRANSPOSED_IN)
        tout = bool(self.mpi_flags & TRANSPOSED_OUT)
        if kind == "r2c" and tin:
            raise ValueError("r2c takes its real input in the normal distribution (reference rdft2-problem.c)")
        if kind == "c2r" and tout:
            raise ValueError("c2r delivers its real output in the normal distribution")
        if kind == "r2r" and (self.r2r_kinds is None or len(self.r2r_kinds) != len(n)):
            raise ValueError("r2r needs one kind per dimension")
        # extents of the array that is exchanged (complex side for real-data transforms)
        self.ne = list(n)
        if kind in ("r2c", "c2r"):
            self.ne[-1] = n[-1] // 2 + 1
        self.nr = list(n)                       # extents of the padded real array
        self.nr[-1] = 2 * (n[-1] // 2 + 1)
        self.b0 = block0 or default_block(self.ne[0], world)
        self.b1 = block1 or default_block(self.ne[1], world)
        (self.alloc_local, self.ln0, self.s0, self.ln1, self.s1) = local_size_many_transposed(
            self.ne, howmany, self.b0, self.b1, world, rank)
        self.R = _prod(self.ne[2:]) * howmany
        self.x = x_local.reshape(-1)
        self.y = y_local.reshape(-1)
        need_x = self.alloc_local * (2 if kind == "r2c" else 1)
        need_y = self.alloc_local * (2 if kind == "c2r" else 1)
        if self.x.numel() < need_x or self.y.numel() < need_y:
            raise ValueError("local arrays need alloc_local = %d elements" % self.alloc_local)
        self.device = self.x.device
        self.edtype = torch.float64 if kind == "r2r" else torch.complex128
        na = max(1, self.alloc_local)
        self.work_a = torch.empty(na, dtype=self.edtype, device=self.device)
        self.work_b = torch.empty(na, dtype=self.edtype, device=self.device)
        self.stages = []          # ("plan", (plan, src, dst)) | ("a2a", _Exchange)
        if kind == "c2r":
            self._build_c2r(tin)
        else:
            self._build(1 if tin else 0, 1 if tout else 0)

    # -- geometry helpers ---------------------------------------------------
    def _cnt(self, axis, r):
        """rows of distributed dim `axis` that rank r owns"""
        return block(self.ne[axis], self.b0 if axis == 0 else self.b1, r)

    def _start(self, axis, r):
        return min(self.ne[axis], r * (self.b0 if axis == 0 else self.b1))

    def _add_plan(self, plan, src, dst):
        self.stages.append(("plan", (plan, src, dst)))

    def _copy_plan(self, loops, src, dst):
        """strided copy as a rank-0 plan (the reference's idiom for rearrangements, fftw/mpi/rearrange.c)"""
        if src.is_complex():
            return fa.plan_guru64_dft([], loops, src, dst, fa.FORWARD, self.flags)
        return fa.plan_guru64_r2r([], loops, src, dst, [], self.flags)

    def _rest_dims(self, ext, unit):
        """guru dims of dims 2.. of a dense array with extents `ext` (+ howmany innermost)"""
        out, stride = [], _prod(ext[2:]) * unit
        for v in ext[2:]:
            stride //= v
            out.append((v, stride))
        return out

    def _local_plan(self, tdims, hd, src, dst, what, dims_idx):
        """tdims: (n, is, os) of the transformed dims; what: 'c2c' | 'r2c' | 'c2r' | 'r2r'"""
        if what == "c2c":
            return fa.plan_guru64_dft(tdims, hd, src, dst, self.sign, self.flags)
        if what == "r2c":
            return fa.plan_guru64_dft_r2c(tdims, hd, src, dst, self.flags)
        if what == "c2r":
            return fa.plan_guru64_dft_c2r(tdims, hd, src, dst, self.flags)
        return fa.plan_guru64_r2r(tdims, hd, src, dst, [self.r2r_kinds[d] for d in dims_idx], self.flags)

    def _switch(self, send, recv, slab, axis):
        """`send` holds [n_other][cnt_me][R] (destination-major); afterwards `slab`
        holds [part_me][n_axis][R]: dim `axis` is local, dim `other` distributed."""
        world, me, R = self.world, self.rank, self.R
        other = 1 - axis
        cnt_me, part_me, N_a = self._cnt(axis, me), self._cnt(other, me), self.ne[axis]
        send_counts = [self._cnt(other, r) * cnt_me * R for r in range(world)]
        recv_counts = [part_me * self._cnt(axis, s) * R for s in range(world)]
        self.stages.append(("a2a", _Exchange(send, recv, send_counts, recv_counts, self.group, world)))
        off = 0
        for s in range(world):
            cnt_s = self._cnt(axis, s)
            if part_me > 0 and cnt_s > 0:
                run = cnt_s * R
                src_blk, dst_blk = recv[off:], slab[self._start(axis, s) * R:]
                self._add_plan(self._copy_plan([(part_me, run, N_a * R), (run, 1, 1)], src_blk, dst_blk),
                               src_blk, dst_blk)
            off += part_me * cnt_s * R

    # -- c2c / r2c / r2r -------------------------------------------------------
    def _build(self, axis, final_axis):
        """axis: distributed dim of the input; final_axis: of the output"""
        ne, R, hm, me = self.ne, self.R, self.howmany, self.rank
        other = 1 - axis
        cnt_me, part_me = self._cnt(axis, me), self._cnt(other, me)
        N_o, N_a = ne[other], ne[axis]
        real_in = (self.kind == "r2c")
        first = "r2c" if real_in else ("r2r" if self.kind == "r2r" else "c2c")
        second = "r2r" if self.kind == "r2r" else "c2c"
        # ---- stage 1: dim `other` and dims 2.. of src [cnt_me][N_o][rest][hm]
        #      -> work_a [N_o][cnt_me][rest][hm]
        if cnt_me > 0:
            if real_in:
                Rr = _prod(self.nr[2:]) * hm
                row_in = self.nr[1] * Rr if len(ne) > 2 else self.nr[1] * hm
                tdims = [(self.n[1], Rr if len(ne) > 2 else hm, cnt_me * R)]
                rin = self._rest_dims(self.nr, hm)
            else:
                row_in = N_o * R
                tdims = [(N_o, R, cnt_me * R)]
                rin = self._rest_dims(ne, hm)
            rout = self._rest_dims(ne, hm)
            for d, ((v, si), (_, so)) in enumerate(zip(rin, rout)):
                tdims.append((self.n[d + 2], si, so))
            hd = [(cnt_me, row_in, R)] + ([(hm, 1, 1)] if hm > 1 else [])
            idx = [other] + list(range(2, len(ne)))
            self._add_plan(self._local_plan(tdims, hd, self.x, self.work_a, first, idx), self.x, self.work_a)
        need_back = (other != final_axis)
        slab = self.work_a if need_back else self.y
        self._switch(self.work_a, self.work_b, slab, axis)
        # ---- stage 2: dim `axis`, slab [part_me][N_a][R]
        if not need_back:
            if part_me > 0:
                hd = [(part_me, N_a * R, N_a * R), (R, 1, 1)]
                self._add_plan(self._local_plan([(N_a, R, R)], hd, slab, slab, second, [axis]), slab, slab)
            return
        if part_me > 0:       # out of place, destination-major: work_b [N_a][part_me][R]
            hd = [(part_me, N_a * R, R), (R, 1, 1)]
            self._add_plan(self._local_plan([(N_a, R, part_me * R)], hd, slab, self.work_b, second, [axis]),
                           slab, self.work_b)
        self._switch(self.work_b, self.work_a, self.y, other)

    # -- c2r --------------------------------------------------------------------
    def _build_c2r(self, tin):
        """backward: dim 0 (complex) first, then the local c2r over dims 1.."""
        ne, R, hm, me = self.ne, self.R, self.howmany, self.rank
        ln0, ln1 = self._cnt(0, me), self._cnt(1, me)
        src = self.x
        if not tin:
            # redistribute [ln0][n1][R] -> [ln1][n0][R]: copy into destination-major order, exchange
            if ln0 > 0:
                loops = [(ne[1], R, ln0 * R), (ln0, ne[1] * R, R), (R, 1, 1)]
                self._add_plan(self._copy_plan(loops, self.x, self.work_a), self.x, self.work_a)
            src = self.x            # the input is destroyed (FFTW_DESTROY_INPUT semantics of c2r)
            self._switch(self.work_a, self.work_b, src, 0)
        # src [ln1][n0][R]: backward complex DFT along dim 0, written destination-major
        if ln1 > 0:
            hd = [(ln1, ne[0] * R, R), (R, 1, 1)]
            p = fa.plan_guru64_dft([(ne[0], R, ln1 * R)], hd, src, self.work_a, fa.BACKWARD, self.flags)
            self._add_plan(p, src, self.work_a)
        self._switch(self.work_a, self.work_b, self.work_a, 1)
        # work_a [ln0][ne1][rest][hm] complex -> y real [ln0][n1][.. 2*nh][hm]
        if ln0 > 0:
            Rr = _prod(self.nr[2:]) * hm
            if len(ne) > 2:
                tdims = [(self.n[1], R, Rr)]
                row_out = self.nr[1] * Rr
            else:
                tdims = [(self.n[1], hm, hm)]
                row_out = self.nr[1] * hm
            for d, ((v, si), (_, so)) in enumerate(zip(self._rest_dims(ne, hm), self._rest_dims(self.nr, hm))):
                tdims.append((self.n[d + 2], si, so))
            hd = [(ln0, ne[1] * R, row_out)] + ([(hm, 1, 1)] if hm > 1 else [])
            self._add_plan(fa.plan_guru64_dft_c2r(tdims, hd, self.work_a, self.y, self.flags),
                           self.work_a, self.y)

    # -- execution ------------------------------------------------------------
    def execute(self):
        """Local plans run on the stream they were bound to at creation (torch's
        current stream) and the collective is ordered after / before them by
        torch.distributed, so the stages execute in program order."""
        for kind, payload in self.stages:
            if kind == "plan":
                plan, src, dst = payload
                self.executor(plan, src, dst)
            else:
                payload.run()

    def sync(self):
        if self.device.type == "cuda":
            import torch
            torch.cuda.synchronize(self.device)

    def num_local_plans(self):
        return sum(1 for k, _ in self.stages if k == "plan")

    def num_exchanges(self):
        return sum(1 for k, _ in self.stages if k == "a2a")


# ---- planners with the reference's names and argument order ----------------

def plan_many_dft(n, howmany, block0, block1, x_local, y_local, sign, flags=fa.ESTIMATE, **kw):
    """fftw_mpi_plan_many_dft(rnk, n, howmany, block, tblock, in, out, comm, sign, flags)"""
    return SlabPlan("c2c", n, howmany, block0, block1, x_local, y_local, sign, flags, **kw)


def plan_dft(n, x_local, y_local, sign, flags=fa.ESTIMATE, **kw):
    return plan_many_dft(n, 1, DEFAULT_BLOCK, DEFAULT_BLOCK, x_local, y_local, sign, flags, **kw)


def plan_dft_2d(n0, n1, x_local, y_local, sign, flags=fa.ESTIMATE, **kw):
    return plan_dft([n0, n1], x_local, y_local, sign, flags, **kw)


def plan_dft_3d(n0, n1, n2, x_local, y_local, sign, flags=fa.ESTIMATE, **kw):
    return plan_dft([n0, n1, n2], x_local, y_local, sign, flags, **kw)


def plan_many_dft_r2c(n, howmany, iblock, oblock, x_local, y_local, flags=fa.ESTIMATE, **kw):
    """fftw_mpi_plan_many_dft_r2c"""
    return SlabPlan("r2c", n, howmany, iblock, oblock, x_local, y_local, fa.FORWARD, flags, **kw)


def plan_dft_r2c(n, x_local, y_local, flags=fa.ESTIMATE, **kw):
    return plan_many_dft_r2c(n, 1, DEFAULT_BLOCK, DEFAULT_BLOCK, x_local, y_local, flags, **kw)


def plan_dft_r2c_2d(n0, n1, x_local, y_local, flags=fa.ESTIMATE, **kw):
    return plan_dft_r2c([n0, n1], x_local, y_local, flags, **kw)


def plan_dft_r2c_3d(n0, n1, n2, x_local, y_local, flags=fa.ESTIMATE, **kw):
    return plan_dft_r2c([n0, n1, n2], x_local, y_local, flags, **kw)


def plan_many_dft_c2r(n, howmany, iblock, oblock, x_local, y_local, flags=fa.ESTIMATE, **kw):
    """fftw_mpi_plan_many_dft_c2r (the complex input is overwritten)"""
    return SlabPlan("c2r", n, howmany, iblock, oblock, x_local, y_local, fa.BACKWARD, flags, **kw)


def plan_dft_c2r(n, x_local, y_local, flags=fa.ESTIMATE, **kw):
    return plan_many_dft_c2r(n, 1, DEFAULT_BLOCK, DEFAULT_BLOCK, x_local, y_local, flags, **kw)


def plan_dft_c2r_2d(n0, n1, x_local, y_local, flags=fa.ESTIMATE, **kw):
    return plan_dft_c2r([n0, n1], x_local, y_local, flags, **kw)


def plan_dft_c2r_3d(n0, n1, n2, x_local, y_local, flags=fa.ESTIMATE, **kw):
    return plan_dft_c2r([n0, n1, n2], x_local, y_local, flags, **kw)


def plan_many_r2r(n, howmany, iblock, oblock, x_local, y_local, kinds, flags=fa.ESTIMATE, **kw):
    """fftw_mpi_plan_many_r2r"""
    return SlabPlan("r2r", n, howmany, iblock, oblock, x_local, y_local, fa.FORWARD, flags,
                    r2r_kinds=kinds, **kw)


def plan_r2r(n, x_local, y_local, kinds, flags=fa.ESTIMATE, **kw):
    return plan_many_r2r(n, 1, DEFAULT_BLOCK, DEFAULT_BLOCK, x_local, y_local, kinds, flags, **kw)


def plan_r2r_2d(n0, n1, x_local, y_local, kind0, kind1, flags=fa.ESTIMATE, **kw):
    return plan_r2r([n0, n1], x_local, y_local, [kind0, kind1], flags, **kw)


def plan_r2r_3d(n0, n1, n2, x_local, y_local, kind0, kind1, kind2, flags=fa.ESTIMATE, **kw):
    return plan_r2r([n0, n1, n2], x_local, y_local, [kind0, kind1, kind2], flags, **kw)
