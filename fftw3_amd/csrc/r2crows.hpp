/*
 * r2crows.hpp -- real rows to half spectra in ONE trip: the half-length complex DFT of
 * the pair sequence z[j] = x[2j] + i x[2j+1] (two register stages, as passrr.hpp) and the
 * r2c untangle, fused through the LDS.  For contiguous real rows of n = 2L, L = R1 R2 in
 * {64, 128, 256, 512, 1024}: the last dimension of 2-D / 3-D real transforms and batched
 * 1-D ones (256^3, 512^3, 1024^2 ...), which otherwise cost a pass plus an untangle trip.
 *
 * A workgroup of 256 items owns T = 4096 / L rows (16 complex elements per item):
 *   A   l = a + R2 i   butterfly g -> (t, a), a fastest     DFT-R1 over i -> d, times w_L^(a d)
 *   x   image E[t][d][a]  (one real plane at a time)
 *   B   butterfly j -> (t, d), d fastest                    DFT-R2 over a -> c,  Z[d + R1 c]
 *   u   Z goes to two LDS planes [t][k]; then one item per Hermitian pair (k, L - k):
 *       Y[k] = E + w_n^k O,  Y[L-k] = conj(E - w_n^k O),  E = (Z[k] + conj Z[L-k]) / 2,
 *       O = -i (Z[k] - conj Z[L-k]) / 2     (r2c_post_kernel in kernels.hip)
 * No predicates: out-of-range butterflies / rows redo the last valid one (pass3g.hpp).
 *
 * Reference counterpart: rdft2 plans whose child is a complex DFT over the (r0, r1) pairs
 * followed by the hc2cfdft codelet (ct_hc2c, fftw/fftw_api.c:5661-5845) -- executed there
 * as two sweeps over the row as well.
 */
#ifndef FA_R2CROWS_HPP
#define FA_R2CROWS_HPP

template <int R1, int R2> struct R2CRGeom {
    static constexpr int L = R1 * R2;
    static constexpr int T = (4096 / L) > 0 ? (4096 / L) : 1;
    static constexpr int NBA = T * R2, NBB = T * R1;
    static constexpr int QA = (NBA + 255) / 256, QB = (NBB + 255) / 256;
    static constexpr int SA = R2 + (R2 % 2 == 0 ? 1 : 0);          /* E: stride of d, odd */
    static constexpr int ST = R1 * SA + ((R1 * SA) % 2 == 0 ? 1 : 0);
    static constexpr int SX = L + 1;                               /* Z planes: row stride */
    static constexpr int EX = T * ST;
    static constexpr int SQ = L / 2 + 1;                           /* DCT-II staging: stride of a residue class */
    static constexpr int fa_max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
    static constexpr int lds_doubles = fa_max3(EX, 2 * T * SX, T * 4 * SQ) + 16;
};

#include "r2r_epi.hpp"

struct R2CRArgs {
    const double *src;      /* real rows: row t at src + t * dis[0] (+ loops), unit stride */
    double *dst;            /* complex rows of L + 1 entries, unit stride (2 doubles); or, with an r2r
                               epilogue, real rows of stride os_k */
    i64 os_k, dst_im;       /* epilogue addressing (epi_store) */
    i64 is_k, src_im;       /* prologue addressing (pro_load, c2r side) */
    i64 rn;                 /* r2r length */
    int r2r, twmul, flags;  /* FFTW_AMD_R2R_POST_* or 0; untangle twiddle = table entry k * twmul */
    int post;               /* c2r rows: FFTW_AMD_R2R_POST_E01 / O01 = store y[2j] = v[j], y[2j+1] = +-v[n-1-j]
                               (reals of stride os_k) instead of the plain real pairs; 0 = plain */
    int pre;                /* r2c rows: FFTW_AMD_R2R_PRE_E10 / O10 / E00 / O00 = gather the real sequence from the
                               user's r2r input (element stride is_k) instead of loading pairs; 0 = plain */
    i64 dn[FFTW_AMD_MAX_DIMS], dis[FFTW_AMD_MAX_DIMS], dos[FFTW_AMD_MAX_DIMS];
    const cplx *wL;         /* w_L^m */
    const cplx *tw_lo;      /* two-level table of w_n^m, n = 2L */
    const cplx *tw_hi;
    i64 ntiles;
    int tw_shift;
    int ndims;
};

/* element m of the real sequence the inner r2c of an r2r transform works on, gathered from
   the user's row (what the PRE_* modes of r2r_kernel write to scratch):
   N = 2L is the inner length, a.rn the r2r length n */
template <class A>
FA_DEV double r2r_pre_elem(const A &a, const double *row, i64 m) {
    const i64 n = a.rn;
    switch (a.pre) {
    case FFTW_AMD_R2R_PRE_E10:
    case FFTW_AMD_R2R_PRE_O10: {
        const i64 si = (m < (n + 1) / 2) ? 2 * m : 2 * n - 1 - 2 * m;
        const double v = row[si * a.is_k];
        return (a.pre == FFTW_AMD_R2R_PRE_O10 && (si & 1)) ? -v : v;
    }
    case FFTW_AMD_R2R_PRE_E00:
        return row[(m < n ? m : 2 * (n - 1) - m) * a.is_k];
    case FFTW_AMD_R2R_PRE_O00:
        if (m >= 1 && m <= n) return row[(m - 1) * a.is_k];
        if (m > n + 1) return -row[(2 * (n + 1) - m - 1) * a.is_k];
        return 0.0;
    default:
        return 0.0;
    }
}

template <int R1, int R2>
__global__ void __launch_bounds__(256, 2)
r2crows_kernel(const R2CRArgs a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    typedef R2CRGeom<R1, R2> G;
    constexpr int L = G::L, T = G::T, QA = G::QA, QB = G::QB, SA = G::SA, ST = G::ST, SX = G::SX;
    const int tid = threadIdx.x;

    i64 blk = (i64)blockIdx.x + (i64)blockIdx.y * gridDim.x;
    i64 tile = blk % a.ntiles;
    i64 rest = blk / a.ntiles;
    i64 soff = 0, doff = 0;
    for (int d = 1; d < a.ndims; ++d) {
        i64 idx = rest % a.dn[d];
        rest /= a.dn[d];
        soff += idx * a.dis[d];
        doff += idx * a.dos[d];
    }
    const i64 t0 = tile * T;
    const int Tcur = (int)((a.dn[0] - t0 < T) ? (a.dn[0] - t0) : T);
    const double *src = a.src + soff + t0 * a.dis[0];
    double *dst = a.dst + doff + t0 * a.dos[0];

    /* ---- stage A */
    cplx x[QA][R1];
    int at[QA], aa[QA];
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        int g = u * 256 + tid;
        const int last = Tcur * R2 - 1;
        g = g < last ? g : last;
        at[u] = g / R2;
        aa[u] = g - at[u] * R2;
        if (a.pre == 0) {
            const double *p = src + (i64)at[u] * a.dis[0] + 2 * aa[u];
#pragma unroll
            for (int i = 0; i < R1; ++i) x[u][i] = *reinterpret_cast<const cplx *>(p + (i64)i * (2 * R2));
        } else if (a.pre == FFTW_AMD_R2R_PRE_E10 || a.pre == FFTW_AMD_R2R_PRE_O10) {
            /* filled below from the LDS staging */
        } else {
            /* the r2r pre-processing as a gather inside the row: v[m] for m = 2j, 2j + 1 */
            const double *row = src + (i64)at[u] * a.dis[0];
#pragma unroll
            for (int i = 0; i < R1; ++i) {
                const i64 j = aa[u] + R2 * i;
                x[u][i] = c_make(r2r_pre_elem(a, row, 2 * j), r2r_pre_elem(a, row, 2 * j + 1));
            }
        }
    }
    if (a.pre == FFTW_AMD_R2R_PRE_E10 || a.pre == FFTW_AMD_R2R_PRE_O10) {
        /* DCT-II / DST-II shuffle v[m] = x[2m] (m < n/2), x[2n-1-2m] (else), n = 2L: the T rows
           are staged in the LDS with coalesced loads, split into the four residue classes of the
           source index mod 4 -- then each of the four streams an item gathers (4j, 4j+2 in the
           first half, 2n-1-4j, 2n-3-4j in the second) is contiguous across the lanes */
        constexpr int SQ = G::SQ, RS = 4 * G::SQ;
        constexpr int N2 = 2 * L;
        {
            /* T * 2L = 8192 reals per tile: 32 independent loads per item, then the LDS writes */
            constexpr int NR = (T * N2) / 256;
            double stg[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int e = r * 256 + tid;
                int t = e / N2;
                const int m = e - t * N2;
                t = t < Tcur - 1 ? t : Tcur - 1;
                stg[r] = src[(i64)t * a.dis[0] + (i64)m * a.is_k];
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int e = r * 256 + tid;
                int t = e / N2;
                const int m = e - t * N2;
                t = t < Tcur - 1 ? t : Tcur - 1;
                plane[t * RS + (m & 3) * SQ + (m >> 2)] = stg[r];
            }
        }
        __syncthreads();
        const double sgn = (a.pre == FFTW_AMD_R2R_PRE_O10) ? -1.0 : 1.0;
#pragma unroll
        for (int u = 0; u < QA; ++u) {
            const double *row = plane + at[u] * RS;
#pragma unroll
            for (int i = 0; i < R1; ++i) {
                const int j = aa[u] + R2 * i;              /* pair index: v[2j], v[2j+1] */
                if (j < L / 2) x[u][i] = c_make(row[j], row[2 * SQ + j]);
                else x[u][i] = c_make(sgn * row[3 * SQ + (L - 1 - j)], sgn * row[SQ + (L - 1 - j)]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        RB<R1>::run(x[u]);
        cplx pw[RB<R1>::bits];
#pragma unroll
        for (int s = 0; s < RB<R1>::bits; ++s) pw[s] = a.wL[(aa[u] << s) % L];
        TwTreeR<R1, RB<R1>::bits - 1, 0, false, true>::run(x[u], pw, c_make(1.0, 0.0));
    }

    /* ---- exchange -> stage B owners (d fastest, then t) */
    cplx y[QB][R2];
    int bd[QB], bt[QB];
#pragma unroll
    for (int v = 0; v < QB; ++v) {
        int j = v * 256 + tid;
        const int last = Tcur * R1 - 1;
        j = j < last ? j : last;
        bt[v] = j / R1;
        bd[v] = j - bt[v] * R1;
    }
#pragma unroll
    for (int u = 0; u < QA; ++u)
#pragma unroll
        for (int d = 0; d < R1; ++d) plane[at[u] * ST + d * SA + aa[u]] = x[u][RB<R1>::slot(d)].x;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int q = 0; q < R2; ++q) y[v][q].x = plane[bt[v] * ST + bd[v] * SA + q];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < QA; ++u)
#pragma unroll
        for (int d = 0; d < R1; ++d) plane[at[u] * ST + d * SA + aa[u]] = x[u][RB<R1>::slot(d)].y;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int q = 0; q < R2; ++q) y[v][q].y = plane[bt[v] * ST + bd[v] * SA + q];
    __syncthreads();

    /* ---- stage B, then Z[t][d + R1 c] into the two planes */
    double *zr = plane, *zi = plane + T * SX;
#pragma unroll
    for (int v = 0; v < QB; ++v) {
        RB<R2>::run(y[v]);
#pragma unroll
        for (int c = 0; c < R2; ++c) {
            const cplx w = y[v][RB<R2>::slot(c)];
            zr[bt[v] * SX + bd[v] + R1 * c] = w.x;
            zi[bt[v] * SX + bd[v] + R1 * c] = w.y;
        }
    }
    __syncthreads();

    /* ---- untangle: one item per pair (k, L - k) of a row, k fastest */
    constexpr int NP = L / 2 + 1;
    const int items = Tcur * NP;
    for (int i = tid; i < items; i += 256) {
        const int t = i / NP, k = i - t * NP;
        const int km = L - k;
        const int kr = (km == L) ? 0 : km;
        const double ar = zr[t * SX + k], ai = zi[t * SX + k];
        const double br = zr[t * SX + kr], bi = zi[t * SX + kr];
        const double er = 0.5 * (ar + br), ei = 0.5 * (ai - bi);
        const double dr = 0.5 * (ar - br), di = 0.5 * (ai + bi);
        const cplx o = c_make(di, -dr);                         /* -i D */
        const cplx p = c_mulc(o, tw2(a.tw_lo, a.tw_hi, a.tw_shift, (i64)k * a.twmul));
        cplx yk = c_make(er + p.x, ei + p.y);
        cplx ym = c_make(er - p.x, -(ei - p.y));
        if (k == 0) { yk.y = 0.0; ym.y = 0.0; }
        if (a.r2r == 0) {
            double *row = dst + (i64)t * a.dos[0];
            *reinterpret_cast<cplx *>(row + 2 * k) = yk;
            if (km != k) *reinterpret_cast<cplx *>(row + 2 * km) = ym;
        } else {
            /* r2r epilogue (R2HC, DHT, DCT-II, DST-II, DCT-I, DST-I): reals of stride os_k */
            const i64 doff = (dst - a.dst) + (i64)t * a.dos[0];
            epi_store(a, doff, k, yk);
            if (km != k) epi_store(a, doff, km, ym);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* the transpose: half spectra (L + 1 complex per row) -> real rows of 2L    */
/* ------------------------------------------------------------------------ */
/*   t   one item per pair (k, L - k):  Z'[k] = E' + i O',  Z'[L-k] = conj(E' - i O'),
 *       E' = Y[k] + conj Y[L-k],  O' = (Y[k] - conj Y[L-k]) w_n^-k   (c2r_pre_kernel; Im Y[0]
 *       and Im Y[L] are ignored) -> two LDS planes [t][k]
 *   A,B the unnormalised BACKWARD complex DFT of length L by the (re, im) swap identity:
 *       stage A reads its inputs from the planes (swapped), stage B stores (Im, Re) as the
 *       real pair (x[2j], x[2j+1])
 */
template <int R1, int R2>
__global__ void __launch_bounds__(256, 2)
c2rrows_kernel(const R2CRArgs a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    typedef R2CRGeom<R1, R2> G;
    constexpr int L = G::L, T = G::T, QA = G::QA, QB = G::QB, SA = G::SA, ST = G::ST, SX = G::SX;
    const int tid = threadIdx.x;

    i64 blk = (i64)blockIdx.x + (i64)blockIdx.y * gridDim.x;
    i64 tile = blk % a.ntiles;
    i64 rest = blk / a.ntiles;
    i64 soff = 0, doff = 0;
    for (int d = 1; d < a.ndims; ++d) {
        i64 idx = rest % a.dn[d];
        rest /= a.dn[d];
        soff += idx * a.dis[d];
        doff += idx * a.dos[d];
    }
    const i64 t0 = tile * T;
    const int Tcur = (int)((a.dn[0] - t0 < T) ? (a.dn[0] - t0) : T);
    const double *src = a.src + soff + t0 * a.dis[0];
    double *dst = a.dst + doff + t0 * a.dos[0];

    /* ---- tangle into the planes (already swapped: zr holds Im Z', zi holds Re Z') */
    double *zr = plane, *zi = plane + T * SX;
    {
        constexpr int NP = L / 2 + 1;
        const int items = Tcur * NP;
        for (int i = tid; i < items; i += 256) {
            const int t = i / NP, k = i - t * NP;
            const int km = L - k;
            cplx yk, ym;
            if (a.r2r == 0) {
                const double *row = src + (i64)t * a.dis[0];
                yk = *reinterpret_cast<const cplx *>(row + 2 * k);
                ym = *reinterpret_cast<const cplx *>(row + 2 * km);
            } else {
                /* r2r prologue (HC2R, DCT-III, DST-III): spectrum entries built from the real input */
                const i64 so = (src - a.src) + (i64)t * a.dis[0];
                yk = pro_load(a, so, k);
                ym = pro_load(a, so, km);
            }
            if (k == 0) { yk.y = 0.0; ym.y = 0.0; }
            const cplx e = c_make(yk.x + ym.x, yk.y - ym.y);
            const cplx dd = c_make(yk.x - ym.x, yk.y + ym.y);
            const cplx o = c_mul(dd, tw2(a.tw_lo, a.tw_hi, a.tw_shift, (i64)k * a.twmul));
            /* Z'[k] = E' + i O' ; Z'[L-k] = conj(E' - i O') */
            const double zkr = e.x - o.y, zki = e.y + o.x;
            const double zmr = e.x + o.y, zmi = -(e.y - o.x);
            zr[t * SX + k] = zki;  zi[t * SX + k] = zkr;
            if (km != k && km != L) { zr[t * SX + km] = zmi;  zi[t * SX + km] = zmr; }
        }
    }
    __syncthreads();

    /* ---- stage A from the planes */
    cplx x[QA][R1];
    int at[QA], aa[QA];
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        int g = u * 256 + tid;
        const int last = Tcur * R2 - 1;
        g = g < last ? g : last;
        at[u] = g / R2;
        aa[u] = g - at[u] * R2;
#pragma unroll
        for (int i = 0; i < R1; ++i)
            x[u][i] = c_make(zr[at[u] * SX + aa[u] + R2 * i], zi[at[u] * SX + aa[u] + R2 * i]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        RB<R1>::run(x[u]);
        cplx pw[RB<R1>::bits];
#pragma unroll
        for (int s = 0; s < RB<R1>::bits; ++s) pw[s] = a.wL[(aa[u] << s) % L];
        TwTreeR<R1, RB<R1>::bits - 1, 0, false, true>::run(x[u], pw, c_make(1.0, 0.0));
    }

    /* ---- exchange -> stage B owners (d fastest, then t) */
    cplx y[QB][R2];
    int bd[QB], bt[QB];
#pragma unroll
    for (int v = 0; v < QB; ++v) {
        int j = v * 256 + tid;
        const int last = Tcur * R1 - 1;
        j = j < last ? j : last;
        bt[v] = j / R1;
        bd[v] = j - bt[v] * R1;
    }
#pragma unroll
    for (int u = 0; u < QA; ++u)
#pragma unroll
        for (int d = 0; d < R1; ++d) plane[at[u] * ST + d * SA + aa[u]] = x[u][RB<R1>::slot(d)].x;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int q = 0; q < R2; ++q) y[v][q].x = plane[bt[v] * ST + bd[v] * SA + q];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < QA; ++u)
#pragma unroll
        for (int d = 0; d < R1; ++d) plane[at[u] * ST + d * SA + aa[u]] = x[u][RB<R1>::slot(d)].y;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int q = 0; q < R2; ++q) y[v][q].y = plane[bt[v] * ST + bd[v] * SA + q];

    /* ---- stage B, store the pair (x[2j], x[2j+1]) = (Im, Re) of the swapped result */
    if (a.post == 0) {
#pragma unroll
        for (int v = 0; v < QB; ++v) {
            RB<R2>::run(y[v]);
            double *p = dst + (i64)bt[v] * a.dos[0] + 2 * bd[v];
#pragma unroll
            for (int c = 0; c < R2; ++c) {
                const cplx w = y[v][RB<R2>::slot(c)];
                *reinterpret_cast<cplx *>(p + (i64)c * (2 * R1)) = c_make(w.y, w.x);
            }
        }
        return;
    }
    /* ---- DCT-III / DST-III output shuffle y[2j] = v[j], y[2j+1] = +-v[n-1-j] (n = 2L): the real
       rows v go to the LDS, then coalesced stores read the two interleaved streams back */
    constexpr int N2 = 2 * L, SV = 2 * L + 2;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v) {
        RB<R2>::run(y[v]);
#pragma unroll
        for (int c = 0; c < R2; ++c) {
            const cplx w = y[v][RB<R2>::slot(c)];
            const int k = bd[v] + R1 * c;
            plane[bt[v] * SV + 2 * k] = w.y;
            plane[bt[v] * SV + 2 * k + 1] = w.x;
        }
    }
    __syncthreads();
    {
        constexpr int NR = (T * N2) / 256;
        const double sgn = (a.post == FFTW_AMD_R2R_POST_O01) ? -1.0 : 1.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int e = r * 256 + tid;
            int t = e / N2;
            const int q = e - t * N2;
            t = t < Tcur - 1 ? t : Tcur - 1;
            const double val = (q & 1) ? sgn * plane[t * SV + (N2 - 1 - (q >> 1))] : plane[t * SV + (q >> 1)];
            dst[(i64)t * a.dos[0] + (i64)q * a.os_k] = val;
        }
    }
}

#endif /* FA_R2CROWS_HPP */
