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
    /* real rows (MODE 1 / 2 of pass3s_kernel): two-level table of w_n^m, n = 2L */
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
    /* FFTW_AMD_F_LO_DFT steps (XROW forms, pass3q.hpp): distance between the rows of a tile, in doubles */
    i64 srs, drs;
};

/* MODE 0: complex rows.
   MODE 1: real rows of n = 2L -> half spectra of L + 1 entries in ONE trip: the rows are read as the pair sequence
           z[j] = x[2j] + i x[2j+1] (the same 16-byte loads), and after stage C every item untangles its own
           outputs, Y[k] = E + w_n^k O with E = (Z[k] + conj Z[L-k]) / 2, O = -i (Z[k] - conj Z[L-k]) / 2
           (hc2cfdft, fftw/fftw_api.c:5831-5845); the partner Z[L-k] comes through the LDS plane (real parts, then
           imaginary parts).  The item that owns k = 0 also stores Y[L].
   MODE 2: the transpose, half spectra -> real rows: stage A builds Z'[l] = E' + i O' from Y[l] and Y[L-l]
           (E' = Y[l] + conj Y[L-l], O' = (Y[l] - conj Y[L-l]) w_n^-l; both read from global memory, the second run
           backwards) with real and imaginary part swapped, and stage C stores (Im, Re): the unnormalised backward
           DFT by the swap identity.
   Reference: ct_hc2c + hc2cfdft / hc2cbdft, fftw/fftw_api.c:5661-5845; r2crows.hpp is the two-stage form.
   XROW (MODE 0, T = 2 or 4; FFTW_AMD_F_LO_DFT steps): the T rows of a tile are the inner tile component -- srs / drs
   apart, the tile index runs over dims[0] -- and a DFT of length T across them (no twiddle) is taken in the same
   registers: every tile is transformed as a 2-D array T x L.  The last trip of a two-dimensional transform whose
   strided axis is T x L0 (planner.c emit_rows_lo_dft; pass3q.hpp is the 512-item form for four rows of 4096). */
/* OUTF >= 0: FFTW_AMD_F_SWAP_OUT (bit 0) and FFTW_AMD_F_NT_OUT (bit 1) of the step as compile-time constants (the XROW
   launcher instantiates the four combinations); -1: read from a.flags */
template <int R1, int MODE = 0, bool XROW = false, int OUTF = -1>
__global__ void __launch_bounds__(256, 2)
pass3s_kernel(const P3SArgs a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    typedef P3SGeom<R1> G;
    constexpr int T = G::T, S1 = G::S1, A2S = G::A2S, SD2 = G::SD2;
    const int tid = threadIdx.x;

    i64 tile, soff, doff, twb_unused;
    fa_block_offsets<false>(a, tile, soff, doff, twb_unused);
    const i64 t0 = XROW ? tile : tile * T;
    const int Tcur = XROW ? T : (int)((a.dn[0] - t0 < T) ? (a.dn[0] - t0) : T);
    const double *src = a.src + soff + t0 * a.dis[0];
    double *dst = a.dst + doff + t0 * a.dos[0];
    const i64 rs_in = XROW ? a.srs : a.dis[0], rs_out = XROW ? a.drs : a.dos[0];   /* row to row inside the tile */

    /* ---- stage A: item a = tid, one radix-R1 butterfly per row */
    cplx x[T][R1];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const double *p = src + (i64)t * rs_in + 2 * tid;
        if (MODE == 2 && t < Tcur) {
            const double *row = src + (i64)t * rs_in;
            const cplx wb = tw2(a.tw_lo, a.tw_hi, a.tw_shift, tid);
#pragma unroll
            for (int i = 0; i < R1; ++i) {
                const int l = tid + 256 * i;
                cplx yk = *reinterpret_cast<const cplx *>(row + 2 * l);
                cplx ym = *reinterpret_cast<const cplx *>(row + 2 * (G::L - l));
                if (l == 0) { yk.y = 0.0; ym.y = 0.0; }
                const cplx e = c_make(yk.x + ym.x, yk.y - ym.y);
                const cplx dd = c_make(yk.x - ym.x, yk.y + ym.y);
                const cplx o = c_mul(dd, i ? c_mul(wb, tw2(a.tw_lo, a.tw_hi, a.tw_shift, 256 * i)) : wb);
                x[t][i] = c_make(e.y + o.x, e.x - o.y);              /* (Im Z', Re Z') */
            }
        } else if (t < Tcur) {
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
    if constexpr (XROW) {
        /* DFT-T across the rows of the tile, element by element */
        static_assert(!XROW || T == 2 || T == 4, "cross-row butterfly: two or four rows per tile");
#pragma unroll
        for (int i = 0; i < R1; ++i) {
            cplx c[T];
#pragma unroll
            for (int t = 0; t < T; ++t) c[t] = x[t][i];
            Bfly<T>::run(c);
#pragma unroll
            for (int t = 0; t < T; ++t) x[t][i] = c[t];
        }
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
    const bool sw = OUTF >= 0 ? (OUTF & 1) != 0 : (a.flags & FFTW_AMD_F_SWAP_OUT) != 0;
    const bool nt_out = OUTF >= 0 ? (OUTF & 2) != 0 : (a.flags & FFTW_AMD_F_NT_OUT) != 0;
    if (MODE == 1) {
        constexpr int L = G::L;
#pragma unroll
        for (int v = 0; v < 2; ++v) RB<16>::run(z[v]);
        cplx pz[2][16];
        /* partner of k = kb + 16 R1 c is L - kb - 16 R1 c (one base per butterfly, constant offsets); only k = 0
           wraps onto itself */
        int pb[2], p0[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int kb = cd1[v] + R1 * cd2[v];
            pb[v] = ct[v] * L + L - kb;
            p0[v] = ct[v] * L + ((L - kb) & (L - 1));
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int c = 0; c < 16; ++c) plane[ct[v] * L + cd1[v] + R1 * cd2[v] + 16 * R1 * c] = z[v][c].x;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int c = 0; c < 16; ++c)
                pz[v][c].x = plane[c ? pb[v] - 16 * R1 * c : p0[v]];
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int c = 0; c < 16; ++c) plane[ct[v] * L + cd1[v] + R1 * cd2[v] + 16 * R1 * c] = z[v][c].y;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int c = 0; c < 16; ++c)
                pz[v][c].y = plane[c ? pb[v] - 16 * R1 * c : p0[v]];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int kb = cd1[v] + R1 * cd2[v];
            const cplx wb = tw2(a.tw_lo, a.tw_hi, a.tw_shift, kb);
            double *row = dst + (i64)ct[v] * rs_out;
            if (ct[v] < Tcur) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const double ar = z[v][c].x, ai = z[v][c].y, br = pz[v][c].x, bi = pz[v][c].y;
                    const double er = 0.5 * (ar + br), ei = 0.5 * (ai - bi);
                    const double dr = 0.5 * (ar - br), di = 0.5 * (ai + bi);
                    const cplx w = c ? c_mul(wb, tw2(a.tw_lo, a.tw_hi, a.tw_shift, 16 * R1 * c)) : wb;
                    const cplx q = c_mulc(c_make(di, -dr), w);
                    cplx yk = c_make(er + q.x, ei + q.y);
                    if (c == 0 && kb == 0) {
                        yk.y = 0.0;
                        *reinterpret_cast<cplx *>(row + 2 * L) = c_make(er - q.x, 0.0);
                    }
                    *reinterpret_cast<cplx *>(row + 2 * (kb + 16 * R1 * c)) = yk;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        RB<16>::run(z[v]);
        if (MODE == 2) {
            if (ct[v] < Tcur) {
                double *p = dst + (i64)ct[v] * rs_out + 2 * (cd1[v] + R1 * cd2[v]);
#pragma unroll
                for (int c = 0; c < 16; ++c) *reinterpret_cast<cplx *>(p + (i64)c * (32 * R1)) = c_make(z[v][c].y, z[v][c].x);
            }
            continue;
        }
        if (ct[v] < Tcur) {
            double *p = dst + (i64)ct[v] * rs_out + 2 * (cd1[v] + R1 * cd2[v]);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                cplx w = z[v][c];
                if (sw) { double s = w.x; w.x = w.y; w.y = s; }
                st_sel(p + (i64)c * (32 * R1), w, nt_out);
            }
        }
    }
}

#endif /* FA_PASS3S_HPP */
