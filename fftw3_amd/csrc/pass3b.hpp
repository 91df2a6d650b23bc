/*
 * pass3b.hpp -- Bluestein's algorithm for a whole row in ONE kernel: a length n whose largest prime factor has
 * no butterfly (1031, 1009, 2 x 509 ...) becomes a cyclic convolution of a smooth length nb >= 2n - 1, and for
 * nb <= 8192 the padded row fits one workgroup:
 *
 *   load   b[l] = x[l] conj(w[l])  (l < n),  0  (n <= l < nb),     w[l] = exp(i pi l^2 / n)
 *   A,B,C  B = DFT_nb(b)                         (the three stages of pass3g.hpp)
 *   mul    B[k] *= K[k],   K = DFT_nb(w[0], w[1..n-1], 0.., w[n-1..1]) / nb
 *   x3     one LDS transposition: stage-C owners -> stage-A owners, real and imaginary part swapped
 *   A,B,C  the backward DFT_nb by the swap identity
 *   store  y[k] = c[k] conj(w[k])  (k < n)
 *
 * -- two trips through the three stages, five LDS exchanges, one trip over HBM (n elements in, n out) where the
 * step-by-step plan makes five trips over arrays of nb ~ 2n elements (copy, two passes, pointwise product, two
 * passes, copy).  Reference: dft_bluestein_apply, fftw/fftw_api.c:1642-1688 (chirp :1598-1640, the same w and the
 * same kernel K with its 1 / nb).
 */
#ifndef FA_PASS3B_HPP
#define FA_PASS3B_HPP

struct BlueArgs {
    const double *src;
    double *dst;
    i64 dn[FFTW_AMD_MAX_DIMS], dis[FFTW_AMD_MAX_DIMS], dos[FFTW_AMD_MAX_DIMS];
    i64 is_l, os_l;         /* element strides of the transform index, in doubles */
    const cplx *wL;         /* w_nb^m */
    const cplx *chirp;      /* w[l], l < n */
    const cplx *kern;       /* K[k], k < nb */
    i64 ntiles;
    int n;
    int ndims, flags;
};

/* stages A, B, C of pass3g.hpp on registers: x (stage-A owners: rows at[], positions aa[] + M i) -> z (stage-C
   owners ct[], cd1[], cd2[]; output c of a butterfly sits in z[.][slot(c)], index cd1 + R1 cd2 + R1 R2 c).
   Ends without a barrier: the caller synchronises before it touches the plane again. */
template <int R1, int R2, int R3, int QA, int QC, int NT = 256>
FA_DEV void p3g_core(cplx (*x)[R1], cplx (*z)[R3], double *plane, const cplx *wL, const int Tcur, const int tid,
                     const int *at, const int *aa, int *ct, int *cd1, int *cd2) {
    typedef P3GGeom<R1, R2, R3, NT> G;
    static_assert(QA == G::QA && QC == G::QC, "geometry");
    constexpr int QB = G::QB;
    constexpr int S1 = G::S1, A2S = G::A2S, SD2 = G::SD2;
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        RB<R1>::run(x[u]);
        cplx pw[RB<R1>::bits];
#pragma unroll
        for (int s = 0; s < RB<R1>::bits; ++s) pw[s] = wL[(aa[u] << s) % G::L];
        TwTreeR<R1, RB<R1>::bits - 1, 0, false, true>::run(x[u], pw, c_make(1.0, 0.0));
    }
    cplx y[QB][R2];
    int ba2[QB], bd1[QB], bt[QB];
#pragma unroll
    for (int v = 0; v < QB; ++v) {
        int h = v * NT + tid;
        const int last = Tcur * R1 * R3 - 1;
        h = h < last ? h : last;
        ba2[v] = h % R3;
        bd1[v] = (h / R3) % R1;
        bt[v] = h / (R3 * R1);
    }
#pragma unroll
    for (int u = 0; u < QA; ++u)
#pragma unroll
        for (int d = 0; d < R1; ++d) plane[(at[u] * R1 + d) * S1 + aa[u]] = x[u][RB<R1>::slot(d)].x;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int i = 0; i < R2; ++i) y[v][i].x = plane[(bt[v] * R1 + bd1[v]) * S1 + ba2[v] + R3 * i];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < QA; ++u)
#pragma unroll
        for (int d = 0; d < R1; ++d) plane[(at[u] * R1 + d) * S1 + aa[u]] = x[u][RB<R1>::slot(d)].y;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int i = 0; i < R2; ++i) y[v][i].y = plane[(bt[v] * R1 + bd1[v]) * S1 + ba2[v] + R3 * i];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v) {
        RB<R2>::run(y[v]);
        cplx pw[RB<R2>::bits];
#pragma unroll
        for (int s = 0; s < RB<R2>::bits; ++s) pw[s] = wL[((ba2[v] << s) * R1) % G::L];
        TwTreeR<R2, RB<R2>::bits - 1, 0, false, true>::run(y[v], pw, c_make(1.0, 0.0));
    }
#pragma unroll
    for (int w = 0; w < QC; ++w) {
        int j = w * NT + tid;
        const int last = Tcur * R1 * R2 - 1;
        j = j < last ? j : last;
        cd1[w] = j % R1;
        cd2[w] = (j / R1) % R2;
        ct[w] = j / (R1 * R2);
    }
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int d = 0; d < R2; ++d)
            plane[(bt[v] * R2 + d) * SD2 + ba2[v] * A2S + bd1[v]] = y[v][RB<R2>::slot(d)].x;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < QC; ++w)
#pragma unroll
        for (int q = 0; q < R3; ++q) z[w][q].x = plane[(ct[w] * R2 + cd2[w]) * SD2 + q * A2S + cd1[w]];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int d = 0; d < R2; ++d)
            plane[(bt[v] * R2 + d) * SD2 + ba2[v] * A2S + bd1[v]] = y[v][RB<R2>::slot(d)].y;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < QC; ++w)
#pragma unroll
        for (int q = 0; q < R3; ++q) z[w][q].y = plane[(ct[w] * R2 + cd2[w]) * SD2 + q * A2S + cd1[w]];
#pragma unroll
    for (int w = 0; w < QC; ++w) RB<R3>::run(z[w]);
}

/* NT = 512 (round 3, kernels_bluew.hip): one row of up to 16384 padded points per workgroup of 512 work-items, i.e.
   lengths n up to 8192 (every prime below 8192) in one kernel */
template <int R1, int R2, int R3, int NT = 256>
__global__ void __launch_bounds__(NT, NT == 256 ? 2 : 1)
blue3g_kernel(const BlueArgs a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    typedef P3GGeom<R1, R2, R3, NT> G;
    constexpr int L = G::L, M = G::M, T = G::T, QA = G::QA, QC = G::QC, KS = R1 * R2;
    const int tid = threadIdx.x;

    i64 tile, soff, doff, twb_unused;
    fa_block_offsets<false>(a, tile, soff, doff, twb_unused);
    const i64 t0 = tile * T;
    const int Tcur = (int)((a.dn[0] - t0 < T) ? (a.dn[0] - t0) : T);
    const double *src = a.src + soff + t0 * a.dis[0];
    double *dst = a.dst + doff + t0 * a.dos[0];
    const int n = a.n;

    cplx x[QA][R1];
    cplx z[QC][R3];
    int at[QA], aa[QA], ct[QC], cd1[QC], cd2[QC];
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        int g = u * NT + tid;
        const int last = Tcur * M - 1;
        g = g < last ? g : last;                      /* beyond the tile: redo the last butterfly */
        at[u] = g / M;
        aa[u] = g - at[u] * M;
        const double *row = src + (i64)at[u] * a.dis[0];
#pragma unroll
        for (int i = 0; i < R1; ++i) {
            const int l = aa[u] + M * i;
            if (M * i < (L + 1) / 2 && l < n) {               /* 2n - 1 <= L: the upper half is padding for every n */
                cplx v = *reinterpret_cast<const cplx *>(row + (i64)l * a.is_l);
                if (a.flags & FFTW_AMD_F_SWAP_IN) { const double s = v.x; v.x = v.y; v.y = s; }
                x[u][i] = c_mulc(v, a.chirp[l]);
            } else {
                x[u][i] = c_make(0.0, 0.0);
            }
            /* 512 work-items have 256 VGPRs each and no second workgroup on the CU: keep the scheduler from
               hoisting all R1 element + chirp loads (8 VGPRs a pair) above the first product */
            if (NT > 256 && (i % 8) == 7) __builtin_amdgcn_sched_barrier(0);
        }
    }
    p3g_core<R1, R2, R3, QA, QC, NT>(x, z, plane, a.wL, Tcur, tid, at, aa, ct, cd1, cd2);

    /* pointwise product with K, then to the stage-A owners with (re, im) swapped: the second trip through the
       stages is the backward transform */
#pragma unroll
    for (int w = 0; w < QC; ++w)
#pragma unroll
        for (int c = 0; c < R3; ++c) {
            const int k = cd1[w] + R1 * cd2[w] + KS * c;
            z[w][RB<R3>::slot(c)] = c_mul(z[w][RB<R3>::slot(c)], a.kern[k]);
            if (NT > 256 && (c % 8) == 7) __builtin_amdgcn_sched_barrier(0);
        }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < QC; ++w)
#pragma unroll
        for (int c = 0; c < R3; ++c) plane[ct[w] * L + cd1[w] + R1 * cd2[w] + KS * c] = z[w][RB<R3>::slot(c)].y;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < QA; ++u)
#pragma unroll
        for (int i = 0; i < R1; ++i) x[u][i].x = plane[at[u] * L + aa[u] + M * i];
    __syncthreads();
#pragma unroll
    for (int w = 0; w < QC; ++w)
#pragma unroll
        for (int c = 0; c < R3; ++c) plane[ct[w] * L + cd1[w] + R1 * cd2[w] + KS * c] = z[w][RB<R3>::slot(c)].x;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < QA; ++u)
#pragma unroll
        for (int i = 0; i < R1; ++i) x[u][i].y = plane[at[u] * L + aa[u] + M * i];
    __syncthreads();

    p3g_core<R1, R2, R3, QA, QC, NT>(x, z, plane, a.wL, Tcur, tid, at, aa, ct, cd1, cd2);

#pragma unroll
    for (int w = 0; w < QC; ++w) {
        double *row = dst + (i64)ct[w] * a.dos[0];
        /* an index the compiler cannot identify with the k of the product above: otherwise it keeps the R3 64-bit
           element offsets computed there alive across the whole second trip (30 spilled register pairs at R3 = 32) */
        int kb = cd1[w] + R1 * cd2[w];
        asm volatile("" : "+v"(kb));
#pragma unroll
        for (int c = 0; c < R3; ++c) {
            const int k = kb + KS * c;
            /* 2n - 1 <= L: outputs from (L + 1) / 2 on are never stored, and the last butterfly's share of them is
               dead code */
            if (KS * c >= (L + 1) / 2) continue;
            if (k < n) {
                const cplx v = z[w][RB<R3>::slot(c)];
                cplx o = c_mulc(c_make(v.y, v.x), a.chirp[k]);
                if (a.flags & FFTW_AMD_F_SWAP_OUT) { const double s = o.x; o.x = o.y; o.y = s; }
                *reinterpret_cast<cplx *>(row + (i64)k * a.os_l) = o;
            }
            if (NT > 256 && (c % 8) == 7) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

#endif /* FA_PASS3B_HPP */
