/*
 * pass3s.hpp -- register-resident three-stage pass for contiguous rows of
 * L = 2048 (8 x 16 x 16), 4096 (16 x 16 x 16) or 8192 (32 x 16 x 16, one row per tile).
 *
 * One workgroup (256 work-items) transforms T = 8192 / L whole rows; every
 * work-item keeps 32 elements in registers through three radix stages and two
 * LDS exchanges:
 *
 *   A   l = a + 256 i       item a = tid, i = 0..R1-1 of every row  -> DFT-R1 -> * w_L^(a d1)
 *   x1  image E1[t][d1][a]  -> item (a2, d1, t) gathers a = a2 + 16 i2
 *   B   DFT-16 over i2      -> * w_256^(a2 d2)
 *   x2  image E2[t][d2][a2][d1] -> item (d1, d2, t) gathers a2 = 0..15
 *   C   DFT-16 over a2      ->  X[d1 + R1 d2 + 16 R1 c]
 *
 * Rows are contiguous on both sides: in stage A the 256 items read 4 KiB runs,
 * in stage C lane order (d1, d2) makes every store instruction of the
 * workgroup one contiguous 4 KiB run.  This is the single-pass alternative to
 * splitting 4096 into 64 x 64 (two trips over HBM); the reference does the
 * same sizes as nested Cooley-Tukey nodes inside one plan (dft-ct-dit/16 over
 * dft-ct-dit/16 over n1_16, fftw/fftw_api.c:2078-2202).
 */
#ifndef FA_PASS3S_HPP
#define FA_PASS3S_HPP

template <int R1> struct P3SGeom {
    static constexpr int L = R1 * 256;
    static constexpr int T = 8192 / L;                 /* rows per tile */
    static constexpr int S1 = 272;                     /* E1 row stride: 256 + 16 */
    static constexpr int A2S = R1 + 1;                 /* E2 stride of a2 */
    static constexpr int SD2 = 16 * (R1 + 1) + (R1 == 8 ? 8 : 0);
    static constexpr int E1 = T * R1 * S1;
    static constexpr int E2 = T * 16 * SD2;
    static constexpr int lds_doubles = (E1 > E2 ? E1 : E2) + 16;
};

struct P3SArgs {
    const double *src;
    double *dst;
    i64 dn[FFTW_AMD_MAX_DIMS], dis[FFTW_AMD_MAX_DIMS], dos[FFTW_AMD_MAX_DIMS];
    const cplx *wL;
    i64 ntiles;
    int ndims, flags;
};

template <int R1>
__global__ void __launch_bounds__(256, 2)
pass3s_kernel(const P3SArgs a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    typedef P3SGeom<R1> G;
    constexpr int T = G::T, S1 = G::S1, A2S = G::A2S, SD2 = G::SD2;
    const int tid = threadIdx.x;

    i64 tile, soff, doff, twb_unused;
    fa_block_offsets<false>(a, tile, soff, doff, twb_unused);
    const i64 t0 = tile * T;
    const int Tcur = (int)((a.dn[0] - t0 < T) ? (a.dn[0] - t0) : T);
    const double *src = a.src + soff + t0 * a.dis[0];
    double *dst = a.dst + doff + t0 * a.dos[0];

    /* ---- stage A: item a = tid, one radix-R1 butterfly per row */
    cplx x[T][R1];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const double *p = src + (i64)t * a.dis[0] + 2 * tid;
        if (t < Tcur) {
            ld_run<R1>(x[t], p, 512, (a.flags & FFTW_AMD_F_NT_IN) != 0);
        } else {
#pragma unroll
            for (int i = 0; i < R1; ++i) x[t][i] = c_make(0.0, 0.0);
        }
    }
    if (a.flags & FFTW_AMD_F_SWAP_IN) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < R1; ++i) { double s = x[t][i].x; x[t][i].x = x[t][i].y; x[t][i].y = s; }
    }
    {
        cplx pw[RB<R1>::bits];
#pragma unroll
        for (int s = 0; s < RB<R1>::bits; ++s) pw[s] = a.wL[tid << s];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            RB<R1>::run(x[t]);
            TwTreeR<R1, RB<R1>::bits - 1, 0, false, true>::run(x[t], pw, c_make(1.0, 0.0));
        }
    }

    /* ---- exchange 1 -> stage B owners (a2 fastest, then d1, then t); 2 butterflies per item */
    int ba2[2], bd1[2], bt[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int h = v * 256 + tid;
        ba2[v] = h % 16;
        bd1[v] = (h / 16) % R1;
        bt[v] = h / (16 * R1);
    }
    cplx y[2][16];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int d = 0; d < R1; ++d) plane[(t * R1 + d) * S1 + tid] = x[t][RB<R1>::slot(d)].x;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int i = 0; i < 16; ++i) y[v][i].x = plane[(bt[v] * R1 + bd1[v]) * S1 + ba2[v] + 16 * i];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int d = 0; d < R1; ++d) plane[(t * R1 + d) * S1 + tid] = x[t][RB<R1>::slot(d)].y;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int i = 0; i < 16; ++i) y[v][i].y = plane[(bt[v] * R1 + bd1[v]) * S1 + ba2[v] + 16 * i];
    __syncthreads();

    /* ---- stage B: DFT-16 over i2, twiddle w_256^(a2 d2) = wL[a2 d2 R1] */
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        RB<16>::run(y[v]);
        cplx pw[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) pw[s] = a.wL[(ba2[v] << s) * R1];
        TwTreeR<16, 3, 0, false, true>::run(y[v], pw, c_make(1.0, 0.0));
    }

    /* ---- exchange 2 -> stage C owners (d1 fastest, then d2, then t) */
    int cd1[2], cd2[2], ct[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int h = v * 256 + tid;
        cd1[v] = h % R1;
        cd2[v] = (h / R1) % 16;
        ct[v] = h / (16 * R1);
    }
    cplx z[2][16];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int d = 0; d < 16; ++d)
            plane[(bt[v] * 16 + d) * SD2 + ba2[v] * A2S + bd1[v]] = y[v][d].x;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int q = 0; q < 16; ++q) z[v][q].x = plane[(ct[v] * 16 + cd2[v]) * SD2 + q * A2S + cd1[v]];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int d = 0; d < 16; ++d)
            plane[(bt[v] * 16 + d) * SD2 + ba2[v] * A2S + bd1[v]] = y[v][d].y;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int q = 0; q < 16; ++q) z[v][q].y = plane[(ct[v] * 16 + cd2[v]) * SD2 + q * A2S + cd1[v]];

    /* ---- stage C: DFT-16 over a2, store X[d1 + R1 d2 + 16 R1 c] */
    const bool sw = (a.flags & FFTW_AMD_F_SWAP_OUT) != 0;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        RB<16>::run(z[v]);
        if (ct[v] < Tcur) {
            double *p = dst + (i64)ct[v] * a.dos[0] + 2 * (cd1[v] + R1 * cd2[v]);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                cplx w = z[v][c];
                if (sw) { double s = w.x; w.x = w.y; w.y = s; }
                st_sel(p + (i64)c * (32 * R1), w, (a.flags & FFTW_AMD_F_NT_OUT) != 0);
            }
        }
    }
}

#endif /* FA_PASS3S_HPP */
