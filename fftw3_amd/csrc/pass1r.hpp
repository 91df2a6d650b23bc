/*
 * pass1r.hpp -- one-stage register kernel for dense rows of a short length R = 2 ... 32:
 * the whole transform is ONE butterfly in the registers of one work-item.
 *
 * This is the counterpart of the reference's leaf executor: one n1_R codelet call covering the
 * whole vector (direct_apply, fftw/fftw_api.c:3130-3205; codelets
 * fftw/dft_scalar/codelets/n1_2.c ... n1_32.c; primes without a codelet run its O(n^2) generic
 * solver, fftw/fftw_api.c:3390-3448).
 *
 * A workgroup of 256 work-items takes T = 256 Q consecutive rows (Q R <= 32 elements per item).
 * The rows are dense (row r starts at element r R), so the tile is one contiguous run of T R
 * elements: every load and store instruction of the workgroup moves one contiguous 4 KiB piece
 * (16 bytes per lane), whatever R is.  The transposition "element e of the run -> position
 * e % R of row e / R" goes through one LDS plane (real parts, then imaginary parts; odd row
 * stride R | 1, so the row-wise accesses of the 64 lanes fall on different banks):
 *
 *   global -> registers (coalesced)  -> LDS [row][pos] -> registers of the row's owner
 *   butterfly R
 *   registers -> LDS [row][pos]      -> registers (coalesced order) -> global
 *
 * Only the global accesses are predicated (rows beyond the end of the batch); the LDS traffic is
 * straight-line code.  Two workgroups fit a CU (plane <= 68 KiB, <= 256 VGPRs).
 */
#ifndef FA_PASS1R_HPP
#define FA_PASS1R_HPP

template <int R> struct P1RGeom {
    static constexpr int Q = R == 4 ? 6 : ((32 / R > 8) ? 8 : 32 / R);   /* rows per work-item (4: 8 rows spill) */
    static constexpr int T = 256 * Q;                         /* rows per tile */
    static constexpr int N = Q * R;                           /* elements per work-item */
    static constexpr int S = R | 1;                           /* LDS row stride (doubles), odd */
    static constexpr int lds_doubles = T * S + 16;
};

template <int R>
__global__ void __launch_bounds__(256, 2)
pass1r_kernel(const P3SArgs a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    typedef P1RGeom<R> G;
    constexpr int Q = G::Q, T = G::T, N = G::N, S = G::S;
    const int tid = threadIdx.x;

    i64 tile, soff, doff, twb_unused;
    fa_block_offsets<false>(a, tile, soff, doff, twb_unused);
    const i64 t0 = tile * T;
    const i64 left = a.dn[0] - t0;
    const int ecur = (int)(left < T ? left : T) * R;          /* valid elements of this tile */
    const double *src = a.src + soff + t0 * (2 * R);
    double *dst = a.dst + doff + t0 * (2 * R);
    const bool nt_in = (a.flags & FFTW_AMD_F_NT_IN) != 0;
    const bool nt_out = (a.flags & FFTW_AMD_F_NT_OUT) != 0;

    /* ---- the run, in coalesced order: item tid holds elements j 256 + tid */
    cplx v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = j * 256 + tid;
        if (e < ecur) v[j] = nt_in ? ld_cplx<true>(src + 2 * e) : ld_cplx<false>(src + 2 * e);
        else v[j] = c_make(0.0, 0.0);
    }
    if (a.flags & FFTW_AMD_F_SWAP_IN) {
#pragma unroll
        for (int j = 0; j < N; ++j) { double s = v[j].x; v[j].x = v[j].y; v[j].y = s; }
    }
    /* position of element e = j 256 + tid in the [row][pos] image: e + (e / R) (S - R); recomputed at each
       use (a multiply-shift) -- kept in an array it costs N registers and pushes R = 4, 29, 31 into scratch */
#define FA_P1R_LP(j) (((j) * 256 + tid) + (((j) * 256 + tid) / R) * (S - R))

    /* ---- rows to their owners: item tid owns rows q 256 + tid */
    cplx x[Q][R];
#pragma unroll
    for (int j = 0; j < N; ++j) plane[FA_P1R_LP(j)] = v[j].x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int k = 0; k < R; ++k) x[q][k].x = plane[(q * 256 + tid) * S + k];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N; ++j) plane[FA_P1R_LP(j)] = v[j].y;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int k = 0; k < R; ++k) x[q][k].y = plane[(q * 256 + tid) * S + k];
    __syncthreads();

#pragma unroll
    for (int q = 0; q < Q; ++q) RB<R>::run(x[q]);

    /* ---- and back */
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int k = 0; k < R; ++k) plane[(q * 256 + tid) * S + k] = x[q][RB<R>::slot(k)].x;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N; ++j) v[j].x = plane[FA_P1R_LP(j)];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int k = 0; k < R; ++k) plane[(q * 256 + tid) * S + k] = x[q][RB<R>::slot(k)].y;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N; ++j) v[j].y = plane[FA_P1R_LP(j)];

    const bool sw = (a.flags & FFTW_AMD_F_SWAP_OUT) != 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = j * 256 + tid;
        cplx w = v[j];
        if (sw) { double s = w.x; w.x = w.y; w.y = s; }
        if (e < ecur) st_sel(dst + 2 * e, w, nt_out);
    }
}

#undef FA_P1R_LP

/* ------------------------------------------------------------------------------------------------------------ */
/* Dense REAL rows of n = 2R = 4 ... 64 points <-> half spectra of R + 1 entries, one work-item per row: the pair   */
/* sequence z[j] = x[2j] + i x[2j+1] is the row itself read as R complex numbers, its DFT is one butterfly, and the  */
/* untangle / tangle (hc2cfdft / hc2cbdft, fftw/fftw_api.c:5831-5845) pairs Z[k] with Z[R-k] inside the item's own   */
/* registers.  The real side is one contiguous run of T R complex slots, the complex side one of T (R + 1): both go  */
/* through the LDS plane in coalesced order, as in pass1r_kernel.  FWD: r2c; else c2r (unnormalised backward).       */
/* ------------------------------------------------------------------------------------------------------------ */
template <int R> struct P1RRealGeom {
    static constexpr int Q = P1RGeom<R>::Q, T = P1RGeom<R>::T;
    static constexpr int NR = Q * R;                          /* real-side complex slots per item */
    static constexpr int NC = Q * (R + 1);                    /* complex-side entries per item */
    static constexpr int S = (R + 1) | 1;                     /* LDS row stride for both images, odd */
    static constexpr int lds_doubles = T * S + 16;
};

/* PAD: the real rows are padded to 2 (R + 1) reals (FFTW's in-place layout): the real side is then a run of R + 1
   slots per row too, the last one padding -- read and ignored (FWD), never written (c2r) */
template <int R, bool FWD, bool PAD = false>
__global__ void __launch_bounds__(256, 2)
pass1r_real_kernel(const P3SArgs a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    typedef P1RRealGeom<R> G;
    constexpr int Q = G::Q, T = G::T, S = G::S;
    constexpr int WI = (FWD && !PAD) ? R : R + 1, WO = (FWD || PAD) ? R + 1 : R;   /* slots per row, load / store side */
    constexpr int NI = Q * WI, NO = Q * WO;
    const int tid = threadIdx.x;

    i64 tile, soff, doff, twb_unused;
    fa_block_offsets<false>(a, tile, soff, doff, twb_unused);
    const i64 t0 = tile * T;
    const i64 left = a.dn[0] - t0;
    const int rows = (int)(left < T ? left : T);
    const double *src = a.src + soff + t0 * (2 * WI);
    double *dst = a.dst + doff + t0 * (2 * WO);
    const int icur = rows * WI, ocur = rows * WO;

    cplx v[NI > NO ? NI : NO];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int e = j * 256 + tid;
        v[j] = e < icur ? *reinterpret_cast<const cplx *>(src + 2 * e) : c_make(0.0, 0.0);
    }
#define FA_P1R_POS(j, W) (((j) * 256 + tid) + (((j) * 256 + tid) / (W)) * (S - (W)))
    cplx x[Q][R + 1];
#pragma unroll
    for (int j = 0; j < NI; ++j) plane[FA_P1R_POS(j, WI)] = v[j].x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int k = 0; k < WI; ++k) x[q][k].x = plane[(q * 256 + tid) * S + k];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) plane[FA_P1R_POS(j, WI)] = v[j].y;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int k = 0; k < WI; ++k) x[q][k].y = plane[(q * 256 + tid) * S + k];
    __syncthreads();

    cplx y[Q][R + 1];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (FWD) {
            cplx z[R];
#pragma unroll
            for (int k = 0; k < R; ++k) z[k] = x[q][k];
            RB<R>::run(z);
            /* Y[k] = E + w_n^k O,  E = (Z[k] + conj Z[R-k]) / 2,  O = -i (Z[k] - conj Z[R-k]) / 2 */
#pragma unroll
            for (int k = 0; k <= R; ++k) {
                const cplx zk = z[RB<R>::slot(k % R)], zm = z[RB<R>::slot((R - k) % R)];
                const double er = 0.5 * (zk.x + zm.x), ei = 0.5 * (zk.y - zm.y);
                const double dr = 0.5 * (zk.x - zm.x), di = 0.5 * (zk.y + zm.y);
                const cplx p = c_mulc(c_make(di, -dr), tw2(a.tw_lo, a.tw_hi, a.tw_shift, k));
                y[q][k] = c_make(er + p.x, ei + p.y);
            }
            y[q][0].y = 0.0;
            y[q][R].y = 0.0;
        } else {
            /* Z'[k] = E' + i O',  E' = Y[k] + conj Y[R-k],  O' = (Y[k] - conj Y[R-k]) w_n^-k; backward DFT by the
               (re, im) swap identity */
            cplx z[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                cplx yk = x[q][k], ym = x[q][R - k];
                if (k == 0) { yk.y = 0.0; ym.y = 0.0; }
                const cplx e = c_make(yk.x + ym.x, yk.y - ym.y);
                const cplx dd = c_make(yk.x - ym.x, yk.y + ym.y);
                const cplx o = c_mul(dd, tw2(a.tw_lo, a.tw_hi, a.tw_shift, k));
                z[k] = c_make(e.y + o.x, e.x - o.y);
            }
            RB<R>::run(z);
#pragma unroll
            for (int k = 0; k < R; ++k) y[q][k] = c_make(z[RB<R>::slot(k)].y, z[RB<R>::slot(k)].x);
            if (PAD) y[q][R] = c_make(0.0, 0.0);
        }
    }

#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int k = 0; k < WO; ++k) plane[(q * 256 + tid) * S + k] = y[q][k].x;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NO; ++j) v[j].x = plane[FA_P1R_POS(j, WO)];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int k = 0; k < WO; ++k) plane[(q * 256 + tid) * S + k] = y[q][k].y;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NO; ++j) v[j].y = plane[FA_P1R_POS(j, WO)];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int e = j * 256 + tid;
        if (e < ocur && (FWD || !PAD || (e % (R + 1)) != R)) *reinterpret_cast<cplx *>(dst + 2 * e) = v[j];
    }
#undef FA_P1R_POS
}

#endif /* FA_PASS1R_HPP */
