/*
 * pass3g.hpp -- register-resident three-stage pass for contiguous rows of a
 * mixed-radix length L = R1 * R2 * R3 <= 4096 (1000 = 10 x 10 x 10, 1536 = 16 x 12 x 8,
 * 3000 = 15 x 20 x 10, ...): the general form of pass3s.hpp (which keeps the tuned
 * 8/16 x 16 x 16 kernels for 2048 / 4096).
 *
 * One workgroup of 256 work-items transforms T whole rows (T * L <= 8192):
 *
 *   A   l = a + M i     (M = R2 R3; butterfly g -> (t, a), a fastest)   DFT-R1 over i -> d1,
 *                        times w_L^(a d1)
 *   x1  image E1[t][d1][a]
 *   B   a = a2 + R3 i2  (butterfly h -> (t, d1, a2), a2 fastest)         DFT-R2 over i2 -> d2,
 *                        times w_M^(a2 d2) = w_L^(a2 d2 R1)
 *   x2  image E2[t][d2][a2][d1]
 *   C   (butterfly j -> (t, d2, d1), d1 fastest)                         DFT-R3 over a2 -> c
 *       X[d1 + R1 d2 + R1 R2 c]
 *
 * Rows are contiguous on both sides: stage A's lanes read consecutive a, stage C's
 * lanes write consecutive d1 + R1 d2 (runs of R1 R2 elements).
 *
 * There are no predicates: a work-item whose butterfly number lies beyond the tile (a
 * partial last slot, or rows past the end of the batch) recomputes the LAST valid
 * butterfly instead -- it loads the same data and stores the same values to the same
 * places, which is harmless, and the straight-line code keeps every array in registers
 * (a run-time `if` around the LDS traffic of passrr.hpp once cost 60-120 spilled VGPRs).
 *
 * Reference counterpart: nested Cooley-Tukey nodes inside one plan, e.g.
 * (dft-ct-dit/10 (dftw-direct-10 "t1_10") (dft-ct-dit/10 ... (dft-direct-10 "n1_10")))
 * for n = 1000 (fftw/fftw_api.c:2078-2202; codelets fftw/dft_scalar/codelets/t1_10.c, n1_10.c).
 */
#ifndef FA_PASS3G_HPP
#define FA_PASS3G_HPP

/* NT = work-items per workgroup: 256 (tiles of 8192 elements, two workgroups per CU) or, for the rows of
   8193 ... 16384 points of r3w_menu.inc (round 3, kernels_r3w.hip), 512 (one row of up to 16384 elements per
   workgroup, one workgroup per CU, like pass3w.hpp) */
constexpr int fa_3g_q(int nb, int nt = 256) { return (nb + nt - 1) / nt; }
/* rows per tile: as many as fit 32 NT elements with at most 32 elements per item in every stage */
constexpr int fa_3g_tile(int R1, int R2, int R3, int nt = 256) {
    const int L = R1 * R2 * R3;
    int T = (32 * nt) / L;
    if (T < 1) T = 1;
    while (T > 1 && (fa_3g_q(T * R2 * R3, nt) * R1 > 32 || fa_3g_q(T * R1 * R3, nt) * R2 > 32 ||
                     fa_3g_q(T * R1 * R2, nt) * R3 > 32)) --T;
    return T;
}

template <int R1, int R2, int R3, int NT = 256> struct P3GGeom {
    static constexpr int L = R1 * R2 * R3;
    static constexpr int M = R2 * R3;
    static constexpr int T = fa_3g_tile(R1, R2, R3, NT);
    static constexpr int NBA = T * M, NBB = T * R1 * R3, NBC = T * R1 * R2;
    static constexpr int QA = fa_3g_q(NBA, NT), QB = fa_3g_q(NBB, NT), QC = fa_3g_q(NBC, NT);
    static constexpr bool fits = QA * R1 <= 32 && QB * R2 <= 32 && QC * R3 <= 32;
    static constexpr int S1 = M + (M % 2 == 0 ? 1 : 0);            /* E1 row stride, odd */
    static constexpr int A2S = R1 + (R1 % 2 == 0 ? 1 : 0);         /* E2 stride of a2, odd */
    static constexpr int SD2 = R3 * A2S + ((R3 * A2S) % 2 == 0 ? 1 : 0);
    static constexpr int E1 = T * R1 * S1;
    static constexpr int E2 = T * R2 * SD2;
    static constexpr int lds_doubles = (E1 > E2 ? E1 : E2) + 16;
};

/* MODE 0: complex rows; MODE 1: real rows of n = 2L -> half spectra; MODE 2: half spectra -> real rows -- the
   fused untangle / tangle of pass3s_kernel (pass3s.hpp) for the general factorisation */
template <int R1, int R2, int R3, int MODE = 0, int NT = 256>
__global__ void __launch_bounds__(NT, NT == 256 ? 2 : 1)
pass3g_kernel(const P3SArgs a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    typedef P3GGeom<R1, R2, R3, NT> G;
    constexpr int M = G::M, T = G::T, QA = G::QA, QB = G::QB, QC = G::QC;
    constexpr int S1 = G::S1, A2S = G::A2S, SD2 = G::SD2;
    const int tid = threadIdx.x;

    i64 tile, soff, doff, twb_unused;
    fa_block_offsets<false>(a, tile, soff, doff, twb_unused);
    const i64 t0 = tile * T;
    const int Tcur = (int)((a.dn[0] - t0 < T) ? (a.dn[0] - t0) : T);
    const double *src = a.src + soff + t0 * a.dis[0];
    double *dst = a.dst + doff + t0 * a.dos[0];

    /* ---- stage A */
    cplx x[QA][R1];
    int at[QA], aa[QA];
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        int g = u * NT + tid;
        const int last = Tcur * M - 1;
        g = g < last ? g : last;                      /* beyond the tile: redo the last butterfly */
        at[u] = g / M;
        aa[u] = g - at[u] * M;
        const double *p = src + (i64)at[u] * a.dis[0] + 2 * aa[u];
        if (MODE == 2) {
            const double *row = src + (i64)at[u] * a.dis[0];
            const cplx wb = tw2(a.tw_lo, a.tw_hi, a.tw_shift, aa[u]);
#pragma unroll
            for (int i = 0; i < R1; ++i) {
                const int l = aa[u] + M * i;
                cplx yk = *reinterpret_cast<const cplx *>(row + 2 * l);
                cplx ym = *reinterpret_cast<const cplx *>(row + 2 * (G::L - l));
                if (l == 0) { yk.y = 0.0; ym.y = 0.0; }
                const cplx e = c_make(yk.x + ym.x, yk.y - ym.y);
                const cplx dd = c_make(yk.x - ym.x, yk.y + ym.y);
                const cplx o = c_mul(dd, i ? c_mul(wb, tw2(a.tw_lo, a.tw_hi, a.tw_shift, M * i)) : wb);
                x[u][i] = c_make(e.y + o.x, e.x - o.y);              /* (Im Z', Re Z') */
            }
        } else {
            ld_run<R1>(x[u], p, (i64)(2 * M), (a.flags & FFTW_AMD_F_NT_IN) != 0);
        }
    }
    if (a.flags & FFTW_AMD_F_SWAP_IN) {
#pragma unroll
        for (int u = 0; u < QA; ++u)
#pragma unroll
            for (int i = 0; i < R1; ++i) { double s = x[u][i].x; x[u][i].x = x[u][i].y; x[u][i].y = s; }
    }
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        RB<R1>::run(x[u]);
        cplx pw[RB<R1>::bits];
#pragma unroll
        for (int s = 0; s < RB<R1>::bits; ++s) pw[s] = a.wL[(aa[u] << s) % G::L];
        TwTreeR<R1, RB<R1>::bits - 1, 0, false, true>::run(x[u], pw, c_make(1.0, 0.0));
    }

    /* ---- exchange 1 -> stage B owners (a2 fastest, then d1, then t) */
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

    /* ---- stage B: DFT-R2 over i2, twiddle w_M^(a2 d2) = wL[a2 d2 R1] */
#pragma unroll
    for (int v = 0; v < QB; ++v) {
        RB<R2>::run(y[v]);
        cplx pw[RB<R2>::bits];
#pragma unroll
        for (int s = 0; s < RB<R2>::bits; ++s) pw[s] = a.wL[((ba2[v] << s) * R1) % G::L];
        TwTreeR<R2, RB<R2>::bits - 1, 0, false, true>::run(y[v], pw, c_make(1.0, 0.0));
    }

    /* ---- exchange 2 -> stage C owners (d1 fastest, then d2, then t) */
    cplx z[QC][R3];
    int cd1[QC], cd2[QC], ct[QC];
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

    /* ---- stage C: DFT-R3 over a2, store X[d1 + R1 d2 + R1 R2 c] */
    const bool sw = (a.flags & FFTW_AMD_F_SWAP_OUT) != 0;
    if (MODE == 1) {
        /* every item untangles its own outputs; the partner of k = kb + R1 R2 c is L - kb - R1 R2 c (only k = 0
           wraps onto itself) and comes through the plane, real parts first */
        constexpr int L = G::L, KS = R1 * R2;
#pragma unroll
        for (int w = 0; w < QC; ++w) RB<R3>::run(z[w]);
        cplx pz[QC][R3];
        int pb[QC], p0[QC];
#pragma unroll
        for (int w = 0; w < QC; ++w) {
            const int kb = cd1[w] + R1 * cd2[w];
            pb[w] = ct[w] * L + L - kb;
            p0[w] = ct[w] * L + (kb ? L - kb : 0);
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < QC; ++w)
#pragma unroll
            for (int c = 0; c < R3; ++c) plane[ct[w] * L + cd1[w] + R1 * cd2[w] + KS * c] = z[w][RB<R3>::slot(c)].x;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < QC; ++w)
#pragma unroll
            for (int c = 0; c < R3; ++c) pz[w][c].x = plane[c ? pb[w] - KS * c : p0[w]];
        __syncthreads();
#pragma unroll
        for (int w = 0; w < QC; ++w)
#pragma unroll
            for (int c = 0; c < R3; ++c) plane[ct[w] * L + cd1[w] + R1 * cd2[w] + KS * c] = z[w][RB<R3>::slot(c)].y;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < QC; ++w)
#pragma unroll
            for (int c = 0; c < R3; ++c) pz[w][c].y = plane[c ? pb[w] - KS * c : p0[w]];
#pragma unroll
        for (int w = 0; w < QC; ++w) {
            const int kb = cd1[w] + R1 * cd2[w];
            const cplx wb = tw2(a.tw_lo, a.tw_hi, a.tw_shift, kb);
            double *row = dst + (i64)ct[w] * a.dos[0];
#pragma unroll
            for (int c = 0; c < R3; ++c) {
                const cplx zk = z[w][RB<R3>::slot(c)];
                const double ar = zk.x, ai = zk.y, br = pz[w][c].x, bi = pz[w][c].y;
                const double er = 0.5 * (ar + br), ei = 0.5 * (ai - bi);
                const double dr = 0.5 * (ar - br), di = 0.5 * (ai + bi);
                const cplx tw = c ? c_mul(wb, tw2(a.tw_lo, a.tw_hi, a.tw_shift, KS * c)) : wb;
                const cplx q = c_mulc(c_make(di, -dr), tw);
                cplx yk = c_make(er + q.x, ei + q.y);
                if (c == 0 && kb == 0) {
                    yk.y = 0.0;
                    *reinterpret_cast<cplx *>(row + 2 * L) = c_make(er - q.x, 0.0);
                }
                *reinterpret_cast<cplx *>(row + 2 * (kb + KS * c)) = yk;
            }
        }
        return;
    }
#pragma unroll
    for (int w = 0; w < QC; ++w) {
        RB<R3>::run(z[w]);
        double *p = dst + (i64)ct[w] * a.dos[0] + 2 * (cd1[w] + R1 * cd2[w]);
        if constexpr (NT > 256 && MODE == 0) {
            /* one branch around the whole run of stores (a branch per store costs spilled VGPRs in the 512-item
               kernels, see pass3q.hpp) */
            if (a.flags & FFTW_AMD_F_NT_OUT) {
#pragma unroll
                for (int c = 0; c < R3; ++c) {
                    cplx v = z[w][RB<R3>::slot(c)];
                    st_cplx<true>(p + (i64)c * (2 * R1 * R2), sw ? c_make(v.y, v.x) : v);
                }
            } else {
#pragma unroll
                for (int c = 0; c < R3; ++c) {
                    cplx v = z[w][RB<R3>::slot(c)];
                    st_cplx<false>(p + (i64)c * (2 * R1 * R2), sw ? c_make(v.y, v.x) : v);
                }
            }
            continue;
        }
#pragma unroll
        for (int c = 0; c < R3; ++c) {
            cplx v = z[w][RB<R3>::slot(c)];
            if (MODE == 2) { *reinterpret_cast<cplx *>(p + (i64)c * (2 * R1 * R2)) = c_make(v.y, v.x); continue; }
            if (sw) { double s = v.x; v.x = v.y; v.y = s; }
            st_sel(p + (i64)c * (2 * R1 * R2), v, (a.flags & FFTW_AMD_F_NT_OUT) != 0);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* The same three stages for the other positions of a plan (L <= 1024, tile   */
/* of T = 8192 / L >= 8 sequences): column passes (sequences fastest on both  */
/* sides, IN_T = OUT_T = true) without twiddle or with the inter-pass twiddle */
/* on the output (HAS_TW = 1), and the last pass of a split (rows in,         */
/* transposed store: IN_T = false, OUT_T = true) without twiddle or with it   */
/* on the input (HAS_TW = 2).  Arguments are those of pass1024 (P1024Args).   */
/* ------------------------------------------------------------------------ */
/* NT = 512 (round 3, kernels_r3tw.hip): tiles of 16384 elements, one workgroup per CU -- the lengths 1025 ... 2048 get
   8 ... 15 sequences per tile (128 ... 240-byte segments) where the 256-item form has 4 ... 7 (64 ... 112 bytes) */
constexpr int fa_3t_tile(int R1, int R2, int R3, int nt) {
    const int L = R1 * R2 * R3;
    int T = (32 * nt) / L;
    if (nt == 256) return T;                                   /* the 256-item menu was chosen with exactly this tile */
    while (T > 1 && (fa_3g_q(T * R2 * R3, nt) * R1 > 40 || fa_3g_q(T * R1 * R3, nt) * R2 > 40 ||
                     fa_3g_q(T * R1 * R2, nt) * R3 > 40)) --T;
    return T;
}
template <int R1, int R2, int R3, int NT = 256> struct P3TGeom {
    static constexpr int L = R1 * R2 * R3;
    static constexpr int M = R2 * R3;
    static constexpr int T = fa_3t_tile(R1, R2, R3, NT);
    static constexpr int NBA = T * M, NBB = T * R1 * R3, NBC = T * R1 * R2;
    static constexpr int QA = fa_3g_q(NBA, NT), QB = fa_3g_q(NBB, NT), QC = fa_3g_q(NBC, NT);
    static constexpr bool fits = T >= 4 && QA * R1 <= 40 && QB * R2 <= 40 && QC * R3 <= 40;   /* the menu keeps only spill-free ones; T = 4 (64-byte segments) only for 2048 */
    static constexpr int TP = T + 1;                               /* odd stride of t where it is not fastest */
    /* column form: element (d1, a, t) at (d1 * M + a) * T + t, rows of d1 padded to stride = T mod 32 */
    static constexpr int S1C = M * T + ((32 + T - (M * T) % 32) % 32);
    /* exchange 2, column form: (d2, a2, d1, t) at d2 * S2C + a2 * A2C + d1 * T + t.  The stage-B owners write with
       t fastest, then a2: with A2C = R1 T a multiple of 16 doubles a quarter wave (8 t x 2 a2) hits 8 bank pairs
       twice, so the 512-item form pads A2C to 8 mod 16 (the 256-item form keeps its 2 x 80 KiB of LDS per CU) */
    static constexpr int A2C = R1 * T + ((NT > 256 && (R1 * T) % 16 == 0 && T == 8) ? 8 : 0);
    static constexpr int S2C = R3 * A2C + ((32 + T - (R3 * A2C) % 32) % 32);
    /* rows form, exchange 1: (t, d1, a) at (t * R1 + d1) * S1R + a.  The stage-B owners read with d1 fastest, then
       a2: a quarter wave is R1 values of d1 x 16 / R1 values of a2, conflict-free when S1R = 16 / R1 mod 16
       (512-item form, R1 a power of two; otherwise the odd stride of P3GGeom) */
    static constexpr int S1R_ODD = M + (M % 2 == 0 ? 1 : 0);
    static constexpr int S1R = (NT > 256 && (R1 == 2 || R1 == 4 || R1 == 8) && M % 16 == 0) ? M + 16 / R1 : S1R_ODD;
    static constexpr int E1 = (R1 * S1C > T * R1 * S1R ? R1 * S1C : T * R1 * S1R);
    static constexpr int E2C = R2 * S2C;
    static constexpr int E2R = R2 * R3 * R1 * TP;
    static constexpr int lds_doubles = (E1 > (E2C > E2R ? E2C : E2R) ? E1 : (E2C > E2R ? E2C : E2R)) + 16;
};

/* RD = 1 (round 3; rows form with the twiddle on the input only, FFTW_AMD_F_REAL_DEC): the LAST trip of a two-trip
   r2c transform of n = L1 x L real points.  Trip 1 was the plain complex pass of length L1 down the columns of the
   real array read as [L1][L / 2] complex pairs: Z[k1][c] = A[k1][2c] + i A[k1][2c + 1], A[k1][j2] the length-L1 real
   DFT of column j2.  This trip takes the rows k1 = 0 ... L1 / 2 (the tile dim, dn[0] = L1 / 2 + 1): it separates
        A[k1][2c]     = (Z[k1][c] + conj Z[L1 - k1][c]) / 2
        A[k1][2c + 1] = (Z[k1][c] - conj Z[L1 - k1][c]) / (2 i)
   while loading (two rows of L / 2 pairs: the mirror lives on the LOAD side, where it is only an address), applies
   the twiddle w_n^(k1 j2), transforms the row (length L) and stores X[k1 + L1 k2] for k2 < L / 2 in place and the
   conjugate of the others at n - (k1 + L1 k2) = (L1 - k1) + L1 (L - 1 - k2) -- the half spectrum 0 ... n / 2, every
   store a run of T consecutive outputs.  Reference counterpart: the rdft2 Cooley-Tukey node over real codelets,
   hc2cf / r2cf (fftw/rdft_scalar/r2cf/hc2cfdft_*.c under ct_hc2c_direct_apply, fftw/fftw_api.c:5831). */
/* FA_P3T_TIMELINE (tests/micro/p3t_timeline.hip only, never in the product): wave 0 of every workgroup leaves
   wall-clock stamps at the phase boundaries in a.dbg[16 * blockIdx.x + k] */
#ifdef FA_P3T_TIMELINE
#define FA_P3T_STAMP(k) do { if (tid == 0) a.dbg[(i64)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define FA_P3T_STAMP(k) do { } while (0)
#endif
template <int R1, int R2, int R3, bool IN_T, int HAS_TW, int NT = 256, int RD = 0>
__global__ void __launch_bounds__(NT, NT == 256 ? 2 : 1)
pass3t_kernel(const P1024Args a) {
    static_assert(RD == 0 || (!IN_T && HAS_TW == 2 && (R2 * R3) % 2 == 0 && R3 % 2 == 0), "real-decimated rows: rows form, twiddle on the input");
    extern __shared__ __attribute__((aligned(16))) double plane[];
    typedef P3TGeom<R1, R2, R3, NT> G;
    constexpr int M = G::M, T = G::T, QA = G::QA, QB = G::QB, QC = G::QC;
    const int tid = threadIdx.x;
#ifdef FA_P3T_TIMELINE
    /* probe only: delay every other workgroup of the first round by a.lo_sh x 0.64 us (are the lock-step phases of a
       one-workgroup-per-CU kernel worth breaking?) */
    if (a.lo_sh > 0 && blockIdx.x < 256 && (blockIdx.x & 8)) {
        const long long t_go = wall_clock64() + 64LL * a.lo_sh;
        while (wall_clock64() < t_go) __builtin_amdgcn_s_sleep(32);
    }
#endif
    FA_P3T_STAMP(0);

    i64 tile, soff, doff, twb;
    fa_block_offsets<true>(a, tile, soff, doff, twb);
    const i64 t0 = tile * T;
    const int Tcur = (int)((a.dn[0] - t0 < T) ? (a.dn[0] - t0) : T);
    const double *src = a.src + soff + t0 * a.dis[0];
    double *dst = a.dst + doff + t0 * a.dos[0];
    const i64 q0 = twb + t0 * a.dtw[0];

    /* ---- stage A: butterfly g -> (t, a); t fastest for column loads, a fastest for rows */
    cplx x[QA][R1];
    int at[QA], aa[QA];
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        int g = u * NT + tid;
        g = g < G::NBA - 1 ? g : G::NBA - 1;
        int t = IN_T ? g % T : g / M;
        aa[u] = IN_T ? g / T : g % M;
        t = t < Tcur - 1 ? t : Tcur - 1;              /* sequences past the end: redo the last one */
        at[u] = t;
    }
    /* the input twiddle's per-item factor w^(q a): its two table loads go out BEFORE the data loads, so that their
       latency hides behind the tile's (they used to start when the data had arrived) */
    cplx twb_in[HAS_TW == 2 ? QA : 1];
    i64 twq[HAS_TW == 2 ? QA : 1];
    if constexpr (HAS_TW == 2) {
#pragma unroll
        for (int u = 0; u < QA; ++u) {
            i64 q = q0 + (i64)at[u] * a.dtw[0];
            if constexpr (!IN_T && M % 64 == 0 && NT % 64 == 0) {
                /* rows form with M a multiple of the wave: a wave's 64 butterflies lie in ONE row, so q and the
                   powers w^(q M 2^s) are wave-uniform -- scalar loads instead of 2 x bits scattered vector loads
                   per butterfly (tests/micro/p3t_timeline.hip: the input twiddle was the longest phase of the tile) */
                q = ((i64)__builtin_amdgcn_readfirstlane((int)(q >> 32)) << 32) |
                    (i64)(unsigned)__builtin_amdgcn_readfirstlane((int)q);
            }
            twq[u] = q;
            twb_in[u] = tw2(a.tw_lo, a.tw_hi, a.tw_shift, q * aa[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        const int t = at[u];
        if constexpr (RD == 1) {
            /* element l = aa + M i of row k1 is A[k1][l]: pair c = l / 2 of the rows k1 and L1 - k1 of Z */
            const i64 k1 = t0 + t, L1 = 2 * (a.dn[0] - 1);
            const i64 km = k1 ? L1 - k1 : 0;
            const double *r1 = a.src + soff + k1 * a.dis[0] + (i64)(aa[u] >> 1) * a.is_l;
            const double *r2 = a.src + soff + km * a.dis[0] + (i64)(aa[u] >> 1) * a.is_l;
            const i64 step = (i64)(M / 2) * a.is_l;
            const bool odd = (aa[u] & 1) != 0;
#pragma unroll
            for (int i = 0; i < R1; ++i) {
                const cplx v1 = *reinterpret_cast<const cplx *>(r1 + i * step);
                const cplx v2 = *reinterpret_cast<const cplx *>(r2 + i * step);
                x[u][i] = odd ? c_make(0.5 * (v1.y + v2.y), 0.5 * (v2.x - v1.x))
                              : c_make(0.5 * (v1.x + v2.x), 0.5 * (v1.y - v2.y));
            }
            continue;
        }
        const double *p = src + (i64)aa[u] * a.is_l + (i64)t * a.dis[0];
        const i64 step = (i64)M * a.is_l;
        ld_run<R1>(x[u], p, step, (a.flags & FFTW_AMD_F_NT_IN) != 0);
    }
#ifdef FA_P3T_TIMELINE
    __builtin_amdgcn_s_waitcnt(0);
    FA_P3T_STAMP(7);                               /* wave 0's loads have arrived */
#endif
    if (a.flags & FFTW_AMD_F_SWAP_IN) {
#pragma unroll
        for (int u = 0; u < QA; ++u)
#pragma unroll
            for (int i = 0; i < R1; ++i) { double s = x[u][i].x; x[u][i].x = x[u][i].y; x[u][i].y = s; }
    }
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        if (HAS_TW == 2) {
            /* conj(w_N^((a + M i) q)) on the input */
            const i64 q = twq[u];
            const cplx base = twb_in[u];
            cplx pw[RB<R1>::bits];
            if constexpr (!IN_T && M % 64 == 0 && NT % 64 == 0) {
                i64 mm[RB<R1>::bits];
#pragma unroll
                for (int s = 0; s < RB<R1>::bits; ++s) mm[s] = (q * M) << s;
                tw2_uniform<RB<R1>::bits>(pw, a.tw_lo, a.tw_hi, a.tw_shift, mm);
            } else {
#pragma unroll
                for (int s = 0; s < RB<R1>::bits; ++s) pw[s] = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (q * M) << s);
            }
            TwTreeR<R1, RB<R1>::bits - 1, 0, true, false>::run(x[u], pw, base);
        }
        RB<R1>::run(x[u]);
        cplx pw[RB<R1>::bits];
#pragma unroll
        for (int s = 0; s < RB<R1>::bits; ++s) pw[s] = a.w1024[(aa[u] << s) % G::L];
        TwTreeR<R1, RB<R1>::bits - 1, 0, false, true>::run(x[u], pw, c_make(1.0, 0.0));
    }

    FA_P3T_STAMP(1);
    /* ---- exchange 1.  Column form: E1[d1][a][t]; rows form: E1[t][d1][a].  Stage B owners:
       column form (t, a2, d1) with t fastest; rows form (d1, a2, t) with d1 fastest. */
    cplx y[QB][R2];
    int ba2[QB], bd1[QB], bt[QB];
#pragma unroll
    for (int v = 0; v < QB; ++v) {
        int h = v * NT + tid;
        h = h < G::NBB - 1 ? h : G::NBB - 1;
        if (IN_T) { bt[v] = h % T; ba2[v] = (h / T) % R3; bd1[v] = h / (T * R3); }
        else      { bd1[v] = h % R1; ba2[v] = (h / R1) % R3; bt[v] = h / (R1 * R3); }
        bt[v] = bt[v] < Tcur - 1 ? bt[v] : Tcur - 1;
    }
#define FA_E1(t, d1, a_) (IN_T ? ((d1) * G::S1C + (a_) * T + (t)) : (((t) * R1 + (d1)) * G::S1R + (a_)))
#pragma unroll
    for (int u = 0; u < QA; ++u)
#pragma unroll
        for (int d = 0; d < R1; ++d) plane[FA_E1(at[u], d, aa[u])] = x[u][RB<R1>::slot(d)].x;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int i = 0; i < R2; ++i) y[v][i].x = plane[FA_E1(bt[v], bd1[v], ba2[v] + R3 * i)];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < QA; ++u)
#pragma unroll
        for (int d = 0; d < R1; ++d) plane[FA_E1(at[u], d, aa[u])] = x[u][RB<R1>::slot(d)].y;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int i = 0; i < R2; ++i) y[v][i].y = plane[FA_E1(bt[v], bd1[v], ba2[v] + R3 * i)];
    __syncthreads();
#undef FA_E1
    FA_P3T_STAMP(2);

    /* ---- stage B: DFT-R2 over i2, twiddle w_M^(a2 d2) = wL[a2 d2 R1] */
#pragma unroll
    for (int v = 0; v < QB; ++v) {
        RB<R2>::run(y[v]);
        cplx pw[RB<R2>::bits];
#pragma unroll
        for (int s = 0; s < RB<R2>::bits; ++s) pw[s] = a.w1024[((ba2[v] << s) * R1) % G::L];
        TwTreeR<R2, RB<R2>::bits - 1, 0, false, true>::run(y[v], pw, c_make(1.0, 0.0));
    }
    FA_P3T_STAMP(3);

    /* ---- exchange 2 -> stage C owners (t fastest, then d1, then d2): transposed / column store.
       Column form: E2[d2][a2][d1][t]; rows form: the same with t padded to an odd stride. */
    cplx z[QC][R3];
    int cd1[QC], cd2[QC], ct[QC];
#pragma unroll
    for (int w = 0; w < QC; ++w) {
        int j = w * NT + tid;
        j = j < G::NBC - 1 ? j : G::NBC - 1;
        ct[w] = j % T;
        ct[w] = ct[w] < Tcur - 1 ? ct[w] : Tcur - 1;
        cd1[w] = (j / T) % R1;
        cd2[w] = j / (T * R1);
    }
#define FA_E2(t, d2, a2, d1) (IN_T ? ((d2) * G::S2C + (a2) * G::A2C + (d1) * T + (t)) \
                                   : ((((d2) * R3 + (a2)) * R1 + (d1)) * G::TP + (t)))
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int d = 0; d < R2; ++d) plane[FA_E2(bt[v], d, ba2[v], bd1[v])] = y[v][RB<R2>::slot(d)].x;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < QC; ++w)
#pragma unroll
        for (int q = 0; q < R3; ++q) z[w][q].x = plane[FA_E2(ct[w], cd2[w], q, cd1[w])];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < QB; ++v)
#pragma unroll
        for (int d = 0; d < R2; ++d) plane[FA_E2(bt[v], d, ba2[v], bd1[v])] = y[v][RB<R2>::slot(d)].y;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < QC; ++w)
#pragma unroll
        for (int q = 0; q < R3; ++q) z[w][q].y = plane[FA_E2(ct[w], cd2[w], q, cd1[w])];
#undef FA_E2
    FA_P3T_STAMP(4);

    /* ---- stage C: DFT-R3 over a2, optional output twiddle, store X[d1 + R1 d2 + R1 R2 c] */
    const bool sw = (a.flags & FFTW_AMD_F_SWAP_OUT) != 0;
#pragma unroll
    for (int w = 0; w < QC; ++w) {
        RB<R3>::run(z[w]);
        const int tc = ct[w];
        const int kb = cd1[w] + R1 * cd2[w];
        if (HAS_TW == 1) {
            const i64 q = q0 + (i64)tc * a.dtw[0];
            cplx base = tw2(a.tw_lo, a.tw_hi, a.tw_shift, q * kb);
            cplx pw[RB<R3>::bits];
#pragma unroll
            for (int s = 0; s < RB<R3>::bits; ++s) pw[s] = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (q * (R1 * R2)) << s);
            TwTreeR<R3, RB<R3>::bits - 1, 0, true, true>::run(z[w], pw, base);
        }
        double *p = dst + (i64)kb * a.os_l + (i64)tc * a.dos[0];
        const i64 step = (i64)(R1 * R2) * a.os_l;
        if constexpr (RD == 1) {
            /* k2 = kb + R1 R2 c: the first R3 / 2 outputs are below L / 2 and stored in place, the others are the
               conjugates of outputs n - k of the rows this trip does not compute */
            const i64 k1 = t0 + tc, L1 = 2 * (a.dn[0] - 1);
            const bool edge = k1 == 0 || 2 * k1 == L1;          /* rows that are their own mirror */
            double *pm = a.dst + doff + (L1 - k1) * a.dos[0] + (i64)(G::L - 1 - kb) * a.os_l;
            /* one branch around all the stores, not one per store (spilled VGPRs in the 512-item kernels) */
            auto stores = [&](auto ntc) {
                constexpr bool NTS = decltype(ntc)::value;
#pragma unroll
                for (int c = 0; c < R3 / 2; ++c) {
                    cplx v = z[w][RB<R3>::slot(c)];
                    if (c == 0 && k1 == 0 && kb == 0) v.y = 0.0;                        /* X[0] */
                    st_cplx<NTS>(p + c * step, v);
                }
                if (!edge) {
#pragma unroll
                    for (int c = R3 / 2; c < R3; ++c) {
                        const cplx v = z[w][RB<R3>::slot(c)];
                        st_cplx<NTS>(pm - c * step, c_make(v.x, -v.y));
                    }
                } else if (k1 == 0 && kb == 0) {
                    st_cplx<NTS>(p + (R3 / 2) * step, c_make(z[w][RB<R3>::slot(R3 / 2)].x, 0.0));   /* X[n / 2] */
                }
            };
            if (a.flags & FFTW_AMD_F_NT_OUT) stores(std::true_type{}); else stores(std::false_type{});
            continue;
        }
        if constexpr (NT > 256) {
            /* one branch around the run of stores, not one per store (spilled VGPRs in the 512-item kernels) */
            if (a.flags & FFTW_AMD_F_NT_OUT) {
#pragma unroll
                for (int c = 0; c < R3; ++c) {
                    cplx v = z[w][RB<R3>::slot(c)];
                    st_cplx<true>(p + c * step, sw ? c_make(v.y, v.x) : v);
                }
            } else {
#pragma unroll
                for (int c = 0; c < R3; ++c) {
                    cplx v = z[w][RB<R3>::slot(c)];
                    st_cplx<false>(p + c * step, sw ? c_make(v.y, v.x) : v);
                }
            }
            continue;
        }
#pragma unroll
        for (int c = 0; c < R3; ++c) {
            cplx v = z[w][RB<R3>::slot(c)];
            if (sw) { double s = v.x; v.x = v.y; v.y = s; }
            st_sel(p + c * step, v, (a.flags & FFTW_AMD_F_NT_OUT) != 0);
        }
    }
#ifdef FA_P3T_TIMELINE
    FA_P3T_STAMP(5);                               /* stores issued */
    __builtin_amdgcn_s_waitcnt(0);
    FA_P3T_STAMP(6);                               /* wave 0's stores acknowledged */
#endif
}

#endif /* FA_PASS3G_HPP */
