/*
 * pass3q.hpp -- the last trip of a 4096 x 4096 two-dimensional transform: FOUR whole rows of 4096 points per
 * workgroup of 512 work-items (32 elements per item, one workgroup per CU like pass3w.hpp), the row transforms
 * AND a radix-4 butterfly across the four rows in the same registers.
 *
 * Why: a strided axis of 4096 points does not fit the 8192-element tiles of the column kernels at 128-byte
 * segments (round 2: rows in one trip + columns as 64 x 64 in two = three trips, 23 % of the roofline).  Split the
 * column transform as 4096 = 1024 x 4 instead (Cooley-Tukey, decimation in time over the row index r = s + 4 r'):
 *
 *   trip 1   for every s in [0, 4) and column c:  Y_s[k'] = sum_r' x[s + 4 r'][c] w_1024^(r' k'),  times w_4096^(s k')
 *            -- the ordinary strided 1024-point pass (pass1024.hpp, 128-byte segments, twiddle on the output),
 *            stored as [k'][s][c]: the four rows that belong together are 256 KiB of contiguous memory
 *   trip 2   (this kernel) tile k': X[k' + 1024 q][kc] = sum_s w_4^(s q) DFT_4096 over c of Y_s[k'][c]
 *
 * i.e. the 2-D DFT of a 4 x 4096 tile: radix 4 over the rows (no twiddle: the dimensions are separable) combined
 * with the first radix-8 column stage into one 32-element register stage, then the 16 x 32 stages of pass3w:
 *
 *   A   item a = tid holds (s, j = a + 512 i), s < 4, i < 8:  DFT-4 over s -> q,  DFT-8 over i -> d1, times w_4096^(a d1)
 *   x1  image E1[D][a],  D = 8 q + d1
 *   B   a = a2 + 32 i2: butterfly h -> (D, a2), a2 fastest:   DFT-16 over i2 -> d2, times w_512^(a2 d2)
 *   x2  image E2[d2][a2][D]
 *   C   item -> (d1, d2, q), d1 fastest:                      DFT-32 over a2 -> c,   X[q][d1 + 8 d2 + 128 c]
 *
 * Every load instruction of the workgroup moves a contiguous 8 KiB piece of one row, every store instruction four
 * contiguous 2 KiB pieces.  Reference counterpart: rank_geq2_apply (fftw/fftw_api.c:4435-4445) runs the two
 * dimensions as two child plans over the whole array; the Cooley-Tukey node split over them is ct_apply_dit
 * (fftw/fftw_api.c:2078-2202).  Planner: emit_rows_lo_dft (planner.c); the step is a PASS with tile_lo_n = 4 and
 * FFTW_AMD_F_LO_DFT.
 */
#ifndef FA_PASS3Q_HPP
#define FA_PASS3Q_HPP

struct P3QGeom {
    static constexpr int RS = 4, RA = 8, R2 = 16, R3 = 32, NT = 512;
    static constexpr int L = RA * R2 * R3, M = R2 * R3;            /* 4096, 512 */
    static constexpr int ND = RS * RA;                             /* 32 values of D */
    static constexpr int S1 = M + 1;                               /* E1 stride of D, odd */
    static constexpr int A2S = ND + 1;                             /* E2 stride of a2, odd */
    static constexpr int SD2 = R3 * A2S + 8;                       /* E2 stride of d2: 8 mod 32, so that the stage-C
                                                                      gather (d1 fastest, then d2) is conflict-free */
    static constexpr int E1 = ND * S1, E2 = R2 * SD2;
    static constexpr int lds_doubles = (E1 > E2 ? E1 : E2) + 16;
};

/* arguments: P3SArgs (pass3s.hpp) -- dims[0] = the tile loop (k'), srs / drs = distance between the four rows of a
   tile on the source / destination side */
/* SW_OUT / NT_OUT (FFTW_AMD_F_SWAP_OUT / NT_OUT of the step) are compile-time: as run-time branches around the 32
   stores they cost 27 spilled VGPRs in the last butterfly -- 108 bytes per lane of scratch traffic, +20 % bytes
   leaving the XCDs per launch (rocprofv3 --pmc: 647 instead of 537 MB per launch of 1024 tiles) */
template <bool SW_OUT, bool NT_OUT>
__global__ void __launch_bounds__(512, 1)
pass3q_kernel(const P3SArgs a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    typedef P3QGeom G;
    constexpr int RS = G::RS, RA = G::RA, R2 = G::R2, R3 = G::R3, NT = G::NT, M = G::M, ND = G::ND;
    constexpr int S1 = G::S1, A2S = G::A2S, SD2 = G::SD2;
    const int tid = threadIdx.x;

    i64 tile, soff, doff, twb_unused;
    fa_block_offsets<false>(a, tile, soff, doff, twb_unused);
    const double *src = a.src + soff + tile * a.dis[0];
    double *dst = a.dst + doff + tile * a.dos[0];

    /* ---- stage A: x[8 s + i] = row s, column tid + 512 i */
    cplx x[ND];
    {
        const bool nt = (a.flags & FFTW_AMD_F_NT_IN) != 0;
#pragma unroll
        for (int s = 0; s < RS; ++s) ld_run<RA>(x + RA * s, src + (i64)s * a.srs + 2 * tid, (i64)(2 * M), nt);
    }
    if (a.flags & FFTW_AMD_F_SWAP_IN) {
#pragma unroll
        for (int i = 0; i < ND; ++i) { double t = x[i].x; x[i].x = x[i].y; x[i].y = t; }
    }
    /* radix 4 across the rows: (s -> q) for every column element */
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        cplx c[4] = { x[i], x[RA + i], x[2 * RA + i], x[3 * RA + i] };
        Bfly<4>::run(c);
        x[i] = c[0]; x[RA + i] = c[1]; x[2 * RA + i] = c[2]; x[3 * RA + i] = c[3];
    }
    {
        /* radix 8 along the row, then w_4096^(a d1): the same twiddle for the four q */
        cplx pw[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) pw[b] = a.wL[tid << b];
#pragma unroll
        for (int q = 0; q < RS; ++q) {
            RB<RA>::run(x + RA * q);
            TwTreeR<RA, 2, 0, false, true>::run(x + RA * q, pw, c_make(1.0, 0.0));
        }
    }

    /* ---- exchange 1 -> stage B owners (a2 fastest, then D): two radix-16 butterflies per item */
    cplx y[2][R2];
    int ba2[2], bD[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int h = v * NT + tid;
        ba2[v] = h % R3;
        bD[v] = h / R3;
    }
#pragma unroll
    for (int D = 0; D < ND; ++D) plane[D * S1 + tid] = x[RA * (D / RA) + RB<RA>::slot(D % RA)].x;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int i = 0; i < R2; ++i) y[v][i].x = plane[bD[v] * S1 + ba2[v] + R3 * i];
    __syncthreads();
#pragma unroll
    for (int D = 0; D < ND; ++D) plane[D * S1 + tid] = x[RA * (D / RA) + RB<RA>::slot(D % RA)].y;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int i = 0; i < R2; ++i) y[v][i].y = plane[bD[v] * S1 + ba2[v] + R3 * i];
    __syncthreads();

    /* ---- stage B: DFT-16 over i2, twiddle w_512^(a2 d2) = wL[8 a2 d2] */
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        RB<R2>::run(y[v]);
        cplx pw[RB<R2>::bits];
#pragma unroll
        for (int s = 0; s < RB<R2>::bits; ++s) pw[s] = a.wL[((ba2[v] << s) * RA) % G::L];
        TwTreeR<R2, RB<R2>::bits - 1, 0, false, true>::run(y[v], pw, c_make(1.0, 0.0));
    }

    /* ---- exchange 2 -> stage C owners: item = d1 + 8 d2 + 128 q (a wave stores 64 consecutive outputs of a row) */
    cplx z[R3];
    const int cd1 = tid % RA, cd2 = (tid / RA) % R2, cq = tid / (RA * R2);
    const int cD = RA * cq + cd1;
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int d = 0; d < R2; ++d) plane[d * SD2 + ba2[v] * A2S + bD[v]] = y[v][RB<R2>::slot(d)].x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R3; ++q) z[q].x = plane[cd2 * SD2 + q * A2S + cD];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int d = 0; d < R2; ++d) plane[d * SD2 + ba2[v] * A2S + bD[v]] = y[v][RB<R2>::slot(d)].y;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R3; ++q) z[q].y = plane[cd2 * SD2 + q * A2S + cD];

    /* ---- stage C: DFT-32 over a2, store X[q][d1 + 8 d2 + 128 c] */
    RB<R3>::run(z);
    double *p = dst + (i64)cq * a.drs + 2 * (cd1 + RA * cd2);
#pragma unroll
    for (int c = 0; c < R3; ++c) {
        cplx v = z[RB<R3>::slot(c)];
        if (SW_OUT) { double t = v.x; v.x = v.y; v.y = t; }
        st_cplx<NT_OUT>(p + (i64)c * (2 * RA * R2), v);
    }
}

#endif /* FA_PASS3Q_HPP */
