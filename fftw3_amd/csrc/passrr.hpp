/*
 * passrr.hpp -- register-resident two-stage pass for L = R1 * R2: the powers of two
 * 64, 128, 256, 512 and mixed-radix pairs such as 143 = 11 x 13 and 105 = 15 x 7.
 *
 * Same design as pass1024.hpp (which is the R1 = R2 = 32 member of the family,
 * kept separately because it is the measured hot kernel): a workgroup of 256
 * work-items owns a tile of 8192 complex doubles = T sequences of length L
 * (T = 8192 / L = 16 .. 128), every work-item keeps 32 elements in registers,
 * the DFT-L is   radix-R1 butterflies  ->  w_L^(a d)  ->  one LDS exchange  ->
 * radix-R2 butterflies, and the tile is T sequences wide so that global access
 * is in segments of at least T * 16 B >= 256 B.
 *
 * With R < 32 a work-item owns Q = 32 / R butterflies per stage:
 *   stage 1: butterfly g = u*256 + tid  (u < Q1)  over l = a + R2 i,   (a, t) from g
 *   stage 2: butterfly h = v*256 + tid  (v < Q2)  over a,  output k = d + R1 c, (d, t) from h
 * The lane -> (a, t) / (d, t) maps put t (column pass) or a / d (row pass)
 * fastest across lanes, exactly like IN_T / OUT_T of pass1024.hpp.
 *
 * Reference counterpart: the same Cooley-Tukey node the reference executes as
 * dftw_direct_apply over a t1_R codelet plus direct_apply over an n1_R codelet
 * (fftw/fftw_api.c:2315-2324, 3182-3205; fftw/dft_scalar/codelets/t1_16.c,
 * n1_16.c, t1_8.c, n1_8.c ...).
 */
#ifndef FA_PASSRR_HPP
#define FA_PASSRR_HPP

/* radix-R butterfly with its output permutation; bits = ceil(log2 R) */
constexpr int fa_ceil_log2(int r) { int b = 0; while ((1 << b) < r) ++b; return b; }
template <int R> struct RB {
    static FA_DEV void run(cplx *x) { Bfly<R>::run(x); }
    static constexpr int slot(int k) { return k; }
    static constexpr int bits = fa_ceil_log2(R);
};
template <> struct RB<32> {
    static FA_DEV void run(cplx *x) { bfly32(x); }
    static constexpr int slot(int k) { return slot32(k); }
    static constexpr int bits = 5;
};

/* x[slot(D)] *= conj(prefix * w^D), D in [0, 2^bits): product tree over the bits of D */
template <int R, int BIT, int D, bool HAVE, bool PERM> struct TwTreeR {
    static FA_DEV void run(cplx *x, const cplx *pw, cplx acc) {
        TwTreeR<R, BIT - 1, D, HAVE, PERM>::run(x, pw, acc);
        cplx nxt = HAVE ? c_mul(acc, pw[BIT]) : pw[BIT];
        TwTreeR<R, BIT - 1, D + (1 << BIT), true, PERM>::run(x, pw, nxt);
    }
};
template <int R, int D, bool HAVE, bool PERM> struct TwTreeR<R, -1, D, HAVE, PERM> {
    static FA_DEV void run(cplx *x, const cplx *, cplx acc) {
        if constexpr (D < R) {                       /* non power-of-two radix: the tree overshoots */
            constexpr int S = PERM ? RB<R>::slot(D) : D;
            if (HAVE) x[S] = c_mulc(x[S], acc);
        }
    }
};

/* sequences per tile: as many as fit 8192 elements while an item keeps at most
   ~34 (stage 1) / ~40 (stage 2) elements in registers */
#ifndef FA_RR_SMALL_ELEMS
#define FA_RR_SMALL_ELEMS 4096
#endif
#ifndef FA_RR_SMALL_MAXL
#define FA_RR_SMALL_MAXL 256
#endif
#ifndef FA_RR_WG3_MAX
#define FA_RR_WG3_MAX 16
#endif
#ifndef FA_RR_SMALL_WGS
#define FA_RR_SMALL_WGS 4
#endif
/* powers of two without a radix-32 stage run 16 elements per item (tile of 4096):
   with 32 the butterflies' temporaries push the 256-VGPR budget into scratch
   (measured: 60-120 spilled VGPRs, passes at ~2 TB/s instead of ~5) */
constexpr bool fa_rr_small(int R1, int R2) {
    return ((R1 & (R1 - 1)) == 0) && ((R2 & (R2 - 1)) == 0) && R1 * R2 <= FA_RR_SMALL_MAXL;
}
/* elements an item may hold in a stage of radix R: powers of two 32; the odd and
   composite butterflies need more temporaries (measured: 33 / 39 elements spill
   150-200 VGPRs), the 3 x 5 radix 15 carries a scratch array (one butterfly per item) */
constexpr int fa_rr_lim(int R, bool first) {
    return R == 32 ? 32 : (R == 15 ? 15 : (first ? 30 : 28));
}
constexpr int fa_rr_q(int R_other, int T) { return (R_other * T + 255) / 256; }
constexpr int fa_rr_tile(int R1, int R2) {
    if (fa_rr_small(R1, R2)) return FA_RR_SMALL_ELEMS / (R1 * R2);
    int T = 8192 / (R1 * R2);
    /* two powers of two (512 = 32 x 16): exactly 32 elements per item in both stages */
    if (((R1 & (R1 - 1)) == 0) && ((R2 & (R2 - 1)) == 0)) return T;
    while (T > 1 && (fa_rr_q(R2, T) * R1 > fa_rr_lim(R1, true) || fa_rr_q(R1, T) * R2 > fa_rr_lim(R2, false))) --T;
    return T;
}

/* workgroups per CU the register budget is compiled for: 4 for the 16-element tiles,
   3 when no item holds more than 16 elements in either stage, else 2 */
constexpr int fa_rr_wgs(int R1, int R2) {
    if (fa_rr_small(R1, R2)) return FA_RR_SMALL_WGS;
    int T = fa_rr_tile(R1, R2);
    return (fa_rr_q(R2, T) * R1 <= FA_RR_WG3_MAX && fa_rr_q(R1, T) * R2 <= FA_RR_WG3_MAX) ? 3 : 2;
}

template <int R1, int R2> struct RRGeom {
    static constexpr int L = R1 * R2;
    static constexpr int T = fa_rr_tile(R1, R2);
    static constexpr int NB1 = R2 * T;               /* radix-R1 butterflies per tile */
    static constexpr int NB2 = R1 * T;               /* radix-R2 butterflies per tile */
    static constexpr int Q1 = (NB1 + 255) / 256;
    static constexpr int Q2 = (NB2 + 255) / 256;
    static constexpr int SDTT = R2 * T + (T < 32 ? T : 0);
    /* LDS image of one real plane: element (d, a, t), padded against bank conflicts */
    template <bool IN_T, bool OUT_T> static FA_DEV int idx(int d, int a, int t) {
        if (IN_T && OUT_T) return d * SDTT + a * T + t;
        if (!IN_T && OUT_T) return d * (T * (R2 + 1)) + t * (R2 + 1) + a;
        if (IN_T && !OUT_T) return a * (T * (R1 + 1)) + t * (R1 + 1) + d;
        return t * (R1 * (R2 + 1)) + d * (R2 + 1) + a;
    }
    static constexpr int lds_doubles = L * T + (R1 > R2 ? R1 : R2) * T + R1 * T + 64;
};

struct PRRTile {
    const double *src;
    double *dst;
    i64 is_l, os_l;
    i64 dis0, dos0;
    i64 dtw0, q0;
    const cplx *wL;
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
    int Tcur;
    int flags;
    int lo_sh;
    i64 lo_is, lo_os;
};

template <int R1, int R2, bool IN_T, bool OUT_T, int HAS_TW>
FA_DEV void prr_tile(const PRRTile &a, double *plane, const int tid) {
    typedef RRGeom<R1, R2> G;
    constexpr int T = G::T, Q1 = G::Q1, Q2 = G::Q2;
    cplx x[Q1][R1];
    int a1[Q1], t1[Q1];
    bool ok1[Q1];

    /* ---- load + stage 1 */
#pragma unroll
    for (int u = 0; u < Q1; ++u) {
        const int g = u * 256 + tid;
        ok1[u] = (G::NB1 % 256 == 0) || g < G::NB1;  /* the last butterfly slot may be empty */
        t1[u] = IN_T ? (g % T) : (g / R2);
        a1[u] = IN_T ? (g / T) : (g % R2);
        const double *p = a.src + (i64)a1[u] * a.is_l + FA_TILE_SOFF(a, t1[u]);
        const i64 step = (i64)R2 * a.is_l;
        if (ok1[u] && (t1[u] >> a.lo_sh) < a.Tcur) {
            ld_run<R1>(x[u], p, step, (a.flags & FFTW_AMD_F_NT_IN) != 0);
        } else {
#pragma unroll
            for (int i = 0; i < R1; ++i) x[u][i] = c_make(0.0, 0.0);
        }
    }
    if (a.flags & FFTW_AMD_F_SWAP_IN) {
#pragma unroll
        for (int u = 0; u < Q1; ++u)
#pragma unroll
            for (int i = 0; i < R1; ++i) { double s = x[u][i].x; x[u][i].x = x[u][i].y; x[u][i].y = s; }
    }
#pragma unroll
    for (int u = 0; u < Q1; ++u) {
        if (HAS_TW == 2) {
            /* conj(w_N^((a + R2 i) q)) on the input */
            const i64 q = a.q0 + (i64)(t1[u] >> a.lo_sh) * a.dtw0;
            cplx base = tw2(a.tw_lo, a.tw_hi, a.tw_shift, q * a1[u]);
            cplx pw[RB<R1>::bits];
#pragma unroll
            for (int s = 0; s < RB<R1>::bits; ++s) pw[s] = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (q * R2) << s);
            TwTreeR<R1, RB<R1>::bits - 1, 0, true, false>::run(x[u], pw, base);
        }
        RB<R1>::run(x[u]);
        {
            cplx pw[RB<R1>::bits];
#pragma unroll
            for (int s = 0; s < RB<R1>::bits; ++s) pw[s] = a.wL[a1[u] << s];
            TwTreeR<R1, RB<R1>::bits - 1, 0, false, true>::run(x[u], pw, c_make(1.0, 0.0));
        }
    }

    /* ---- exchange, one real plane at a time */
    cplx y[Q2][R2];
    int d2[Q2], t2[Q2];
    bool ok2[Q2];
#pragma unroll
    for (int v = 0; v < Q2; ++v) {
        const int h = v * 256 + tid;
        ok2[v] = (G::NB2 % 256 == 0) || h < G::NB2;
        t2[v] = OUT_T ? (h % T) : (h / R1);
        d2[v] = OUT_T ? (h / T) : (h % R1);
    }
#pragma unroll
    for (int u = 0; u < Q1; ++u)
        if (ok1[u]) {
#pragma unroll
            for (int d = 0; d < R1; ++d) plane[G::template idx<IN_T, OUT_T>(d, a1[u], t1[u])] = x[u][RB<R1>::slot(d)].x;
        }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < Q2; ++v)
#pragma unroll
        for (int q = 0; q < R2; ++q) y[v][q].x = ok2[v] ? plane[G::template idx<IN_T, OUT_T>(d2[v], q, t2[v])] : 0.0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < Q1; ++u)
        if (ok1[u]) {
#pragma unroll
            for (int d = 0; d < R1; ++d) plane[G::template idx<IN_T, OUT_T>(d, a1[u], t1[u])] = x[u][RB<R1>::slot(d)].y;
        }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < Q2; ++v)
#pragma unroll
        for (int q = 0; q < R2; ++q) y[v][q].y = ok2[v] ? plane[G::template idx<IN_T, OUT_T>(d2[v], q, t2[v])] : 0.0;

    /* ---- stage 2, output twiddle, store: X[d + R1 c] */
#pragma unroll
    for (int v = 0; v < Q2; ++v) {
        RB<R2>::run(y[v]);
        if (HAS_TW == 1) {
            const i64 q = a.q0 + (i64)(t2[v] >> a.lo_sh) * a.dtw0;
            cplx base = tw2(a.tw_lo, a.tw_hi, a.tw_shift, q * d2[v]);
            cplx pw[RB<R2>::bits];
#pragma unroll
            for (int s = 0; s < RB<R2>::bits; ++s) pw[s] = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (q * R1) << s);
            TwTreeR<R2, RB<R2>::bits - 1, 0, true, true>::run(y[v], pw, base);
        }
        if (ok2[v] && (t2[v] >> a.lo_sh) < a.Tcur) {
            double *p = a.dst + (i64)d2[v] * a.os_l + FA_TILE_DOFF(a, t2[v]);
            const i64 step = (i64)R1 * a.os_l;
            const bool sw = (a.flags & FFTW_AMD_F_SWAP_OUT) != 0;
            const bool nt_out = (a.flags & FFTW_AMD_F_NT_OUT) != 0;
#pragma unroll
            for (int c = 0; c < R2; ++c) {
                cplx w = y[v][RB<R2>::slot(c)];
                if (sw) { double s = w.x; w.x = w.y; w.y = s; }
                st_sel(p + c * step, w, nt_out);
            }
        }
    }
}

/* kernel arguments are those of pass1024 (P1024Args): same dims / strides / tables */
template <int R1, int R2, bool IN_T, bool OUT_T, int HAS_TW>
__global__ void __launch_bounds__(256, fa_rr_wgs(R1, R2))
passrr_kernel(const P1024Args a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    constexpr int T = RRGeom<R1, R2>::T;
    i64 tile, soff, doff, twb;
    fa_block_offsets<true>(a, tile, soff, doff, twb);
    const i64 t0 = tile * (T >> a.lo_sh);
    PRRTile t;
    t.lo_sh = a.lo_sh; t.lo_is = a.lo_is; t.lo_os = a.lo_os;
    t.src = a.src + soff + t0 * a.dis[0];
    t.dst = a.dst + doff + t0 * a.dos[0];
    t.is_l = a.is_l; t.os_l = a.os_l;
    t.dis0 = a.dis[0]; t.dos0 = a.dos[0];
    t.dtw0 = a.dtw[0]; t.q0 = twb + t0 * a.dtw[0];
    t.wL = a.w1024; t.tw_lo = a.tw_lo; t.tw_hi = a.tw_hi; t.tw_shift = a.tw_shift;
    t.Tcur = (int)((a.dn[0] - t0 < (T >> a.lo_sh)) ? (a.dn[0] - t0) : (T >> a.lo_sh));
    t.flags = a.flags;
    prr_tile<R1, R2, IN_T, OUT_T, HAS_TW>(t, plane, threadIdx.x);
}

#endif /* FA_PASSRR_HPP */
