/*
 * stream1024.hpp -- persistent, software-pipelined form of the 1024-point pass.
 *
 * pass1024.hpp runs one tile per workgroup: load -> compute -> store, so a
 * tile's HBM latency is covered only by the one other workgroup on the CU.
 * Here one workgroup per CU (256 work-items, one wave per SIMD, the whole
 * 512-entry register file) walks a list of tiles and keeps THREE tiles in
 * registers: while tile k is being transformed (x -> LDS -> y), the loads of
 * tiles k+1 and k+2 are already in flight into the other register sets, and
 * the stores of tile k-1 drain.  The exchange is one pass through LDS with
 * 16-byte elements (8448 x 16 B = 132 KiB, conflict-free for ds_write_b128 /
 * ds_read_b128, see lds_index in pass1024.hpp).
 *
 * The tile list can mix the two passes of an N = 1024 x 1024 transform in one
 * launch ("fused" mode): tiles are taken in ticket order
 *     ... pass-1 tiles of transform b,  pass-2 tiles of transform b - LAG ...
 * A pass-2 tile waits until all 128 pass-1 tiles of its transform have
 * published (agent-scope release/acquire, MI355X guide section 6 G16); because
 * a ticket only ever waits for LOWER tickets, and lower tickets are held by
 * workgroups that are already running, the scheme cannot deadlock whatever the
 * dispatch order.  The intermediate of a transform is read back a few
 * microseconds after it was written, i.e. out of the Infinity Cache, which is
 * what lifts the two-pass algorithm above the "two full HBM round trips" bound.
 */
#ifndef FA_STREAM1024_HPP
#define FA_STREAM1024_HPP

#define FA_S1024_LDS_CPLX 8448

struct S1024Pass {
    const double *src;
    double *dst;
    i64 is_l, os_l;
    i64 dn[FFTW_AMD_MAX_DIMS], dis[FFTW_AMD_MAX_DIMS], dos[FFTW_AMD_MAX_DIMS], dtw[FFTW_AMD_MAX_DIMS];
    i64 ntiles0;     /* tiles along dims[0] */
    i64 total;       /* tiles in this pass */
    int ndims, flags;
};

struct S1024Args {
    S1024Pass ps;
    const cplx *w1024;
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
};

struct TileCtx {
    i64 soff, doff, q0;   /* q0: twiddle index of sequence 0 of the tile */
    int Tcur;
};

FA_DEV TileCtx s1024_decode(const S1024Pass &p, i64 tile) {
    TileCtx c;
    i64 t = tile % p.ntiles0;
    i64 rest = tile / p.ntiles0;
    c.soff = 0; c.doff = 0; c.q0 = 0;
    for (int d = 1; d < p.ndims; ++d) {
        i64 idx = rest % p.dn[d];
        rest /= p.dn[d];
        c.soff += idx * p.dis[d];
        c.doff += idx * p.dos[d];
        c.q0 += idx * p.dtw[d];
    }
    i64 t0 = t * 8;
    c.soff += t0 * p.dis[0];
    c.doff += t0 * p.dos[0];
    c.q0 += t0 * p.dtw[0];
    c.Tcur = (int)((p.dn[0] - t0 < 8) ? (p.dn[0] - t0) : 8);
    return c;
}

template <bool IN_T>
FA_DEV void s1024_issue_loads(cplx (&x)[32], const S1024Pass &p, i64 tile, int tid) {
    const int ti = IN_T ? (tid & 7) : (tid >> 5);
    const int ai = IN_T ? (tid >> 3) : (tid & 31);
    TileCtx c = s1024_decode(p, tile);
    const double *ptr = p.src + c.soff + (i64)ai * p.is_l + (i64)ti * p.dis[0];
    const i64 step = 32 * p.is_l;
    if (ti < c.Tcur) {
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = *reinterpret_cast<const cplx *>(ptr + i * step);
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = c_make(0.0, 0.0);
    }
}

/* transform the tile held in x; prefetch tile `next` into x once x is dead */
template <bool IN_T, bool OUT_T, int TW>
FA_DEV void s1024_body(cplx (&x)[32], const S1024Args &a, i64 tile, i64 next, cplx *ex, int tid) {
    const S1024Pass &p = a.ps;
    const int ti = IN_T ? (tid & 7) : (tid >> 5);
    const int ai = IN_T ? (tid >> 3) : (tid & 31);
    const int to = OUT_T ? (tid & 7) : (tid >> 5);
    const int dq = OUT_T ? (tid >> 3) : (tid & 31);
    TileCtx c = s1024_decode(p, tile);

    if (p.flags & FFTW_AMD_F_SWAP_IN) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { double s = x[i].x; x[i].x = x[i].y; x[i].y = s; }
    }
    if (TW == 2) {
        const i64 q = c.q0 + (i64)ti * p.dtw[0];
        cplx base = tw2(a.tw_lo, a.tw_hi, a.tw_shift, q * ai);
        cplx pw[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) pw[s] = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (q * 32) << s);
        TwTree<4, 0, true, false>::run(x, pw, base);
    }
    bfly32(x);
    {
        cplx pw[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) pw[s] = a.w1024[ai << s];
        TwTree<4, 0, false, true>::run(x, pw, c_make(1.0, 0.0));
    }

    /* exchange: every item writes its 32 outputs, then reads the 32 inputs of
       its second butterfly */
#pragma unroll
    for (int d = 0; d < 32; ++d) ex[lds_index<IN_T, OUT_T>(d, ai, ti)] = x[slot32(d)];
    __syncthreads();

    /* x is dead: start the loads of the tile two steps ahead */
    if (next < p.total) s1024_issue_loads<IN_T>(x, p, next, tid);

    cplx y[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) y[q] = ex[lds_index<IN_T, OUT_T>(dq, q, to)];
    __syncthreads();

    bfly32(y);
    if (TW == 1) {
        const i64 q = c.q0 + (i64)to * p.dtw[0];
        cplx base = tw2(a.tw_lo, a.tw_hi, a.tw_shift, q * dq);
        cplx pw[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) pw[s] = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (q * 32) << s);
        TwTree<4, 0, true, true>::run(y, pw, base);
    }
    if (to < c.Tcur) {
        double *ptr = p.dst + c.doff + (i64)dq * p.os_l + (i64)to * p.dos[0];
        const i64 step = 32 * p.os_l;
        const bool sw = (p.flags & FFTW_AMD_F_SWAP_OUT) != 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            cplx v = y[slot32(k)];
            if (sw) { double s = v.x; v.x = v.y; v.y = s; }
            *reinterpret_cast<cplx *>(ptr + k * step) = v;
        }
    }
}

template <bool IN_T, bool OUT_T, int TW>
__global__ void __launch_bounds__(256, 1)
stream1024_kernel(const S1024Args a) {
    extern __shared__ __attribute__((aligned(16))) cplx s1024_ex[];
    const int tid = threadIdx.x;
    const i64 G = gridDim.x;
    const i64 total = a.ps.total;
    i64 tA = blockIdx.x, tB = tA + G;
    cplx xa[32], xb[32];
    if (tA < total) s1024_issue_loads<IN_T>(xa, a.ps, tA, tid);
    if (tB < total) s1024_issue_loads<IN_T>(xb, a.ps, tB, tid);
    for (;;) {
        if (tA >= total) break;
        s1024_body<IN_T, OUT_T, TW>(xa, a, tA, tA + 2 * G, s1024_ex, tid);
        tA += 2 * G;
        if (tB >= total) break;
        s1024_body<IN_T, OUT_T, TW>(xb, a, tB, tB + 2 * G, s1024_ex, tid);
        tB += 2 * G;
    }
}

#endif /* FA_STREAM1024_HPP */
