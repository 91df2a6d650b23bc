/*
 * mixed1024.hpp -- software pipeline of the two passes of N = 1024 x 1024
 * transforms with NO in-kernel synchronisation.
 *
 * Launch k carries the pass-1 tiles of chunk k and the pass-2 tiles of chunk
 * k-1, interleaved block by block.  Pass 2 therefore only ever reads what the
 * PREVIOUS launch wrote (ordinary stream order makes it visible), yet the
 * intermediate of a chunk is consumed one launch (tens of microseconds) after
 * it was produced, while it still sits in the Infinity Cache, and HBM reads
 * (pass 1) and HBM writes (pass 2) are mixed in every launch.  Compare
 * fused1024.hpp, which reaches the same ordering inside one launch but pays for
 * tickets, polls and write-through hand-offs.
 */
#ifndef FA_MIXED1024_HPP
#define FA_MIXED1024_HPP

struct Mixed1024Args {
    const double *in;        /* first transform of chunk k */
    double *out;             /* first transform of chunk k-1 */
    double *slot_w;          /* scratch slot written by pass 1 (chunk k) */
    const double *slot_r;    /* scratch slot read by pass 2 (chunk k-1) */
    i64 in_bs, out_bs;
    int n1, n2;              /* transforms in chunk k (pass 1) and in chunk k-1 (pass 2); 0 = none */
    const cplx *w1024;
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
    int flags;
};

__global__ void __launch_bounds__(256, 2)
mixed1024_kernel(const Mixed1024Args a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    const i64 N1 = 1024;
    const int tid = threadIdx.x;
    /* block -> (kind, index): alternate while both kinds remain, then the rest of the longer one */
    const int c1 = a.n1 * 128, c2 = a.n2 * 128;
    const int both = 2 * (c1 < c2 ? c1 : c2);
    int b = blockIdx.x, kind, idx;
    if (b < both) { kind = b & 1; idx = b >> 1; }
    else { kind = c1 > c2 ? 0 : 1; idx = (both >> 1) + (b - both); }
    const int xf = idx >> 7, tile = idx & 127;

    P1024Tile t;
    t.w1024 = a.w1024; t.tw_lo = a.tw_lo; t.tw_hi = a.tw_hi; t.tw_shift = a.tw_shift;
    t.Tcur = 8;
    t.lo_sh = 0; t.lo_is = 0; t.lo_os = 0;
    if (kind == 0) {
        t.src = a.in + (i64)xf * a.in_bs + (i64)tile * 16;
        t.dst = a.slot_w + (i64)xf * (2 * N1 * N1) + (i64)tile * 16;
        t.is_l = 2 * N1; t.os_l = 2 * N1;
        t.dis0 = 2; t.dos0 = 2;
        t.dtw0 = 0; t.q0 = 0;
        t.flags = a.flags & FFTW_AMD_F_SWAP_IN;
        p1024_tile<true, true, 0>(t, plane, tid);
    } else {
        t.src = a.slot_r + (i64)xf * (2 * N1 * N1) + (i64)tile * 8 * 2 * N1;
        t.dst = a.out + (i64)xf * a.out_bs + (i64)tile * 16;
        t.is_l = 2; t.os_l = 2 * N1;
        t.dis0 = 2 * N1; t.dos0 = 2;
        t.dtw0 = 1; t.q0 = (i64)tile * 8;
        t.flags = a.flags & FFTW_AMD_F_SWAP_OUT;
        p1024_tile<false, true, 2>(t, plane, tid);
    }
}

#endif /* FA_MIXED1024_HPP */
