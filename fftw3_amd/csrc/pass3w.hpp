/*
 * pass3w.hpp -- rows of 16384 points in one trip: the three register stages of pass3g.hpp (32 x 16 x 32) run by
 * ONE workgroup of 512 work-items per row, 32 elements per item, one workgroup per CU (the row's 256 KiB fill half
 * of the CU's register file; a real plane of 128 KiB goes through the LDS at a time).
 *
 *   A   l = a + 512 i      item a = tid, i = 0..31                 DFT-32 over i -> d1, times w_L^(a d1)
 *   x1  image E1[d1][a]
 *   B   a = a2 + 32 i2     butterfly h -> (d1, a2), a2 fastest     DFT-16 over i2 -> d2, times w_512^(a2 d2)
 *   x2  image E2[d2][a2][d1]
 *   C   butterfly j -> (d2, d1), d1 fastest                        DFT-32 over a2 -> c,  X[d1 + 32 d2 + 512 c]
 *
 * One workgroup per CU means no second workgroup covers the barriers and the load / store phases, so the kernel
 * runs below the two-per-CU kernels (measured in profiles/r02_rows_16384.txt) -- but one trip instead of the two of
 * the 128 x 128 plan.  Like the other rows longer than 4096 it has no LDS-kernel fallback (planner: long_rows_ok).
 * Reference counterpart: nested Cooley-Tukey nodes inside one plan (fftw/fftw_api.c:2078-2202).
 */
#ifndef FA_PASS3W_HPP
#define FA_PASS3W_HPP

struct P3WGeom {
    static constexpr int R1 = 32, R2 = 16, R3 = 32, NT = 512;
    static constexpr int L = R1 * R2 * R3, M = R2 * R3;
    static constexpr int S1 = M + 1;                               /* E1 row stride, odd */
    static constexpr int A2S = R1 + 1;                             /* E2 stride of a2, odd */
    static constexpr int SD2 = R3 * A2S + 1;                       /* E2 stride of d2, even + 1 */
    static constexpr int E1 = R1 * S1, E2 = R2 * SD2;
    static constexpr int lds_doubles = (E1 > E2 ? E1 : E2) + 16;
};

/* MODE 0: complex rows; MODE 1: real rows of 32768 -> half spectra; MODE 2: half spectra -> real rows (the fused
   untangle / tangle of pass3s_kernel, pass3s.hpp) */
/* SW_OUT / NT_OUT (MODE 0: FFTW_AMD_F_SWAP_OUT / NT_OUT of the step) are compile-time: as run-time branches around
   the 32 stores they cost 20 spilled VGPRs in the last butterfly (see pass3q.hpp) */
template <int MODE, bool SW_OUT = false, bool NT_OUT = false>
__global__ void __launch_bounds__(512, 1)
pass3w_kernel(const P3SArgs a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    typedef P3WGeom G;
    constexpr int R1 = G::R1, R2 = G::R2, R3 = G::R3, NT = G::NT, M = G::M;
    constexpr int S1 = G::S1, A2S = G::A2S, SD2 = G::SD2;
    const int tid = threadIdx.x;

    i64 tile, soff, doff, twb_unused;
    fa_block_offsets<false>(a, tile, soff, doff, twb_unused);
    const double *src = a.src + soff + tile * a.dis[0];
    double *dst = a.dst + doff + tile * a.dos[0];

    /* ---- stage A: one radix-32 butterfly per item */
    cplx x[R1];
    if (MODE == 2) {
        const cplx wb = tw2(a.tw_lo, a.tw_hi, a.tw_shift, tid);
#pragma unroll
        for (int i = 0; i < R1; ++i) {
            const int l = tid + M * i;
            cplx yk = *reinterpret_cast<const cplx *>(src + 2 * l);
            cplx ym = *reinterpret_cast<const cplx *>(src + 2 * (G::L - l));
            if (l == 0) { yk.y = 0.0; ym.y = 0.0; }
            const cplx e = c_make(yk.x + ym.x, yk.y - ym.y);
            const cplx dd = c_make(yk.x - ym.x, yk.y + ym.y);
            const cplx o = c_mul(dd, i ? c_mul(wb, tw2(a.tw_lo, a.tw_hi, a.tw_shift, M * i)) : wb);
            x[i] = c_make(e.y + o.x, e.x - o.y);                     /* (Im Z', Re Z') */
        }
    } else {
        ld_run<R1>(x, src + 2 * tid, (i64)(2 * M), (a.flags & FFTW_AMD_F_NT_IN) != 0);
    }
    if (a.flags & FFTW_AMD_F_SWAP_IN) {
#pragma unroll
        for (int i = 0; i < R1; ++i) { double s = x[i].x; x[i].x = x[i].y; x[i].y = s; }
    }
    {
        RB<R1>::run(x);
        cplx pw[RB<R1>::bits];
#pragma unroll
        for (int s = 0; s < RB<R1>::bits; ++s) pw[s] = a.wL[(tid << s) % G::L];
        TwTreeR<R1, RB<R1>::bits - 1, 0, false, true>::run(x, pw, c_make(1.0, 0.0));
    }

    /* ---- exchange 1 -> stage B owners (a2 fastest, then d1): two radix-16 butterflies per item */
    cplx y[2][R2];
    int ba2[2], bd1[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int h = v * NT + tid;
        ba2[v] = h % R3;
        bd1[v] = h / R3;
    }
#pragma unroll
    for (int d = 0; d < R1; ++d) plane[d * S1 + tid] = x[RB<R1>::slot(d)].x;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int i = 0; i < R2; ++i) y[v][i].x = plane[bd1[v] * S1 + ba2[v] + R3 * i];
    __syncthreads();
#pragma unroll
    for (int d = 0; d < R1; ++d) plane[d * S1 + tid] = x[RB<R1>::slot(d)].y;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int i = 0; i < R2; ++i) y[v][i].y = plane[bd1[v] * S1 + ba2[v] + R3 * i];
    __syncthreads();

    /* ---- stage B: DFT-16 over i2, twiddle w_512^(a2 d2) = wL[a2 d2 R1] */
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        RB<R2>::run(y[v]);
        cplx pw[RB<R2>::bits];
#pragma unroll
        for (int s = 0; s < RB<R2>::bits; ++s) pw[s] = a.wL[((ba2[v] << s) * R1) % G::L];
        TwTreeR<R2, RB<R2>::bits - 1, 0, false, true>::run(y[v], pw, c_make(1.0, 0.0));
    }

    /* ---- exchange 2 -> stage C owners (d1 fastest, then d2): one radix-32 butterfly per item */
    cplx z[R3];
    const int cd1 = tid % R1, cd2 = tid / R1;
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int d = 0; d < R2; ++d) plane[d * SD2 + ba2[v] * A2S + bd1[v]] = y[v][RB<R2>::slot(d)].x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R3; ++q) z[q].x = plane[cd2 * SD2 + q * A2S + cd1];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int d = 0; d < R2; ++d) plane[d * SD2 + ba2[v] * A2S + bd1[v]] = y[v][RB<R2>::slot(d)].y;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R3; ++q) z[q].y = plane[cd2 * SD2 + q * A2S + cd1];

    /* ---- stage C: DFT-32 over a2, store X[d1 + 32 d2 + 512 c] */
    RB<R3>::run(z);
    if (MODE == 1) {
        /* every item untangles its own 32 outputs; the partner of k = kb + 512 c is L - kb - 512 c (only k = 0
           wraps onto itself) and comes through the plane, real parts first */
        constexpr int L = G::L, KS = R1 * R2;
        const int kb = cd1 + R1 * cd2;
        const int pb = L - kb, p0 = kb ? L - kb : 0;
        cplx pz[R3];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < R3; ++c) plane[kb + KS * c] = z[RB<R3>::slot(c)].x;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < R3; ++c) pz[c].x = plane[c ? pb - KS * c : p0];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < R3; ++c) plane[kb + KS * c] = z[RB<R3>::slot(c)].y;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < R3; ++c) pz[c].y = plane[c ? pb - KS * c : p0];
        const cplx wb = tw2(a.tw_lo, a.tw_hi, a.tw_shift, kb);
#pragma unroll
        for (int c = 0; c < R3; ++c) {
            const cplx zk = z[RB<R3>::slot(c)];
            const double ar = zk.x, ai = zk.y, br = pz[c].x, bi = pz[c].y;
            const double er = 0.5 * (ar + br), ei = 0.5 * (ai - bi);
            const double dr = 0.5 * (ar - br), di = 0.5 * (ai + bi);
            const cplx tw = c ? c_mul(wb, tw2(a.tw_lo, a.tw_hi, a.tw_shift, KS * c)) : wb;
            const cplx q = c_mulc(c_make(di, -dr), tw);
            cplx yk = c_make(er + q.x, ei + q.y);
            if (c == 0 && kb == 0) {
                yk.y = 0.0;
                *reinterpret_cast<cplx *>(dst + 2 * L) = c_make(er - q.x, 0.0);
            }
            *reinterpret_cast<cplx *>(dst + 2 * (kb + KS * c)) = yk;
        }
        return;
    }
    double *p = dst + 2 * (cd1 + R1 * cd2);
#pragma unroll
    for (int c = 0; c < R3; ++c) {
        cplx v = z[RB<R3>::slot(c)];
        if (MODE == 2) { *reinterpret_cast<cplx *>(p + (i64)c * (2 * R1 * R2)) = c_make(v.y, v.x); continue; }
        if (SW_OUT) { double s = v.x; v.x = v.y; v.y = s; }
        st_cplx<NT_OUT>(p + (i64)c * (2 * R1 * R2), v);
    }
}

#endif /* FA_PASS3W_HPP */
