/*
 * kernels.hip -- hand-written HIP kernels for gfx950 (MI355X) and their
 * launchers.  No rocFFT/hipFFT, no MFMA: batched FFT passes are HBM-bound
 * vector-FMA work.
 *
 * What each kernel replaces in the reference (fftw/fftw_api.c, "A.c"):
 *   pass_generic_kernel  <- the leaf executors and the Cooley-Tukey twiddle
 *       step: vrank_geq1_apply A.c:4627, ct_apply_dit A.c:2078, direct_apply
 *       A.c:3182, dftw_direct_apply A.c:2315, dftw_generic A.c:2730-2903 and
 *       the large-radix dftw_genericbuf A.c:2905-3109 (two-level twiddles,
 *       A.c:18920-18941), with the n1/t1 codelets replaced by butterflies.h.
 *   copy_kernel          <- cpy2d_pair copies A.c:16412-16452, Bluestein
 *       chirp products A.c:1642-1688, Rader gather/scatter A.c:4187-4261.
 *   r2c_post / c2r_pre   <- ct_hc2c_direct_apply A.c:5831-5845 with the
 *       hc2cfdft / hc2cbdft codelets, plus the DC/Nyquist zeroing A.c:7155.
 */
#include <string.h>
#include "common.hpp"

#include "pass1024.hpp"

/* ------------------------------------------------------------------------ */
/* generic LDS pass kernel (runtime radices)                                 */
/* ------------------------------------------------------------------------ */

struct PassArgs {
    const double *src;
    double *dst;
    i64 src_im, dst_im;
    i64 is_l, os_l;
    i64 dn[FFTW_AMD_MAX_DIMS], dis[FFTW_AMD_MAX_DIMS], dos[FFTW_AMD_MAX_DIMS], dtw[FFTW_AMD_MAX_DIMS];
    const cplx *wL;
    const cplx *tw_lo;
    const cplx *tw_hi;
    i64 tw_n;
    i64 ntiles;
    int tw_shift;
    int L, nrad;
    int rad[FFTW_AMD_MAX_RADICES];
    int ndims, T, ld, flags;
    int in_t_fast, out_t_fast;
    int lo_n;              /* inner tile component (1: none); T counts lo_n * hi entries */
    i64 lo_is, lo_os;
};

template <int R>
FA_DEV void stockham_stage(const cplx *A, cplx *B, const cplx *wL, int L, int ld, int Tcur,
                           int Ns, int tid, int nth) {
    const int m = L / R;
    const int nb = m * Tcur;
    const int twstep = L / (Ns * R);
    for (int b = tid; b < nb; b += nth) {
        int j = b / Tcur, t = b - j * Tcur;
        int k = j % Ns;
        cplx x[R];
#pragma unroll
        for (int i = 0; i < R; ++i) x[i] = A[(j + i * m) * ld + t];
        if (Ns > 1) {
#pragma unroll
            for (int i = 1; i < R; ++i) x[i] = c_mulc(x[i], wL[i * k * twstep]);
        }
        Bfly<R>::run(x);
        int o = (j - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) B[(o + q * Ns) * ld + t] = x[q];
    }
}

/* any prime radix p: O(p^2) DFT straight out of LDS (reference generic_apply,
   A.c:3428-3448, plays this role for primes without a codelet) */
FA_DEV void stockham_stage_prime(const cplx *A, cplx *B, const cplx *wL, int L, int ld, int Tcur,
                                 int Ns, int p, int tid, int nth) {
    const int m = L / p;
    const int nb = m * Tcur;
    const int twstep = L / (Ns * p);
    const int pstep = L / p;
    for (int b = tid; b < nb; b += nth) {
        int j = b / Tcur, t = b - j * Tcur;
        int k = j % Ns;
        int o = (j - k) * p + k;
        for (int q = 0; q < p; ++q) {
            cplx acc = c_make(0.0, 0.0);
            int iq = 0;
            for (int i = 0; i < p; ++i) {
                cplx xi = A[(j + i * m) * ld + t];
                if (Ns > 1) xi = c_mulc(xi, wL[i * k * twstep]);
                xi = c_mulc(xi, wL[iq * pstep]);
                acc = c_add(acc, xi);
                iq += q;
                if (iq >= p) iq -= p;
            }
            B[(o + q * Ns) * ld + t] = acc;
        }
    }
}

template <bool VIN, bool VOUT>
__global__ void __launch_bounds__(256)
pass_generic_kernel(const PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fa_lds_raw[];
    cplx *A = reinterpret_cast<cplx *>(fa_lds_raw);
    cplx *B = A + (size_t)a.L * a.ld;

    const int tid = threadIdx.x, nth = blockDim.x;
    i64 blk = (i64)blockIdx.x + (i64)blockIdx.y * gridDim.x;
    i64 tile = blk % a.ntiles;
    i64 rest = blk / a.ntiles;
    i64 soff = 0, doff = 0, twb = 0;
    for (int d = 1; d < a.ndims; ++d) {
        i64 idx = rest % a.dn[d];
        rest /= a.dn[d];
        soff += idx * a.dis[d];
        doff += idx * a.dos[d];
        twb += idx * a.dtw[d];
    }
    const int Thi = a.T / a.lo_n;
    const i64 t0 = tile * Thi;
    const int Tcur = (int)((a.dn[0] - t0 < Thi) ? (a.dn[0] - t0) : Thi) * a.lo_n;
    const int L = a.L, ld = a.ld;
    const int total = L * Tcur;
    const bool tw_in = a.tw_n && (a.flags & FFTW_AMD_F_TW_IN);

    /* ---- load tile into LDS, coalesced along whichever index is contiguous */
    for (int e = tid; e < total; e += nth) {
        int l, t;
        if (a.in_t_fast) { l = e / Tcur; t = e - l * Tcur; }
        else             { t = e / L;    l = e - t * L; }
        const int thi = t / a.lo_n, tlo = t - thi * a.lo_n;
        i64 addr = soff + (i64)l * a.is_l + (t0 + thi) * a.dis[0] + tlo * a.lo_is;
        cplx v = load_elem<VIN>(a.src, addr, a.src_im, a.flags);
        if (tw_in) {
            i64 m = (i64)l * (twb + (t0 + thi) * a.dtw[0]);
            v = c_mulc(v, tw2(a.tw_lo, a.tw_hi, a.tw_shift, m));
        }
        A[l * ld + t] = v;
    }
    __syncthreads();

    /* ---- Stockham autosort stages, ping-pong between the two LDS images */
    int Ns = 1;
    for (int s = 0; s < a.nrad; ++s) {
        const int r = a.rad[s];
        switch (r) {
        case 2:  stockham_stage<2>(A, B, a.wL, L, ld, Tcur, Ns, tid, nth); break;
        case 3:  stockham_stage<3>(A, B, a.wL, L, ld, Tcur, Ns, tid, nth); break;
        case 4:  stockham_stage<4>(A, B, a.wL, L, ld, Tcur, Ns, tid, nth); break;
        case 5:  stockham_stage<5>(A, B, a.wL, L, ld, Tcur, Ns, tid, nth); break;
        case 7:  stockham_stage<7>(A, B, a.wL, L, ld, Tcur, Ns, tid, nth); break;
        case 8:  stockham_stage<8>(A, B, a.wL, L, ld, Tcur, Ns, tid, nth); break;
        case 11: stockham_stage<11>(A, B, a.wL, L, ld, Tcur, Ns, tid, nth); break;
        case 13: stockham_stage<13>(A, B, a.wL, L, ld, Tcur, Ns, tid, nth); break;
        case 16: stockham_stage<16>(A, B, a.wL, L, ld, Tcur, Ns, tid, nth); break;
        default: stockham_stage_prime(A, B, a.wL, L, ld, Tcur, Ns, r, tid, nth); break;
        }
        Ns *= r;
        cplx *tmp = A; A = B; B = tmp;
        __syncthreads();
    }

    /* ---- inter-pass twiddle (conj: forward) and store */
    for (int e = tid; e < total; e += nth) {
        int l, t;
        if (a.out_t_fast) { l = e / Tcur; t = e - l * Tcur; }
        else              { t = e / L;    l = e - t * L; }
        cplx v = A[l * ld + t];
        const int thi = t / a.lo_n, tlo = t - thi * a.lo_n;
        if (a.tw_n && !tw_in) {
            i64 m = (i64)l * (twb + (t0 + thi) * a.dtw[0]);
            v = c_mulc(v, tw2(a.tw_lo, a.tw_hi, a.tw_shift, m));
        }
        i64 addr = doff + (i64)l * a.os_l + (t0 + thi) * a.dos[0] + tlo * a.lo_os;
        store_elem<VOUT>(a.dst, addr, a.dst_im, a.flags, v);
    }
}


/* ------------------------------------------------------------------------ */
/* generic LDS pass kernel, in-place form: ONE LDS image (64 KiB for a        */
/* 4096-element tile) so that two workgroups share a CU.  A stage reads all  */
/* of an item's butterflies into registers, synchronises, and writes them    */
/* back to their autosort positions.  Radices with a register butterfly only */
/* (2,3,4,5,7,8,11,13,16); tiles with a larger prime stage use the ping-pong */
/* kernel above.                                                             */
/* ------------------------------------------------------------------------ */
template <int R>
FA_DEV void inplace_stage(cplx *A, const cplx *wL, int L, int ld, int Tcur, int Ns, int tid) {
    constexpr int NB = (4096 / R + 255) / 256;      /* butterflies per item, worst case */
    const int m = L / R;
    const int nb = m * Tcur;
    const int twstep = L / (Ns * R);
    cplx x[NB][R];
    int pos[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int b = u * 256 + tid;
        pos[u] = -1;
        if (b < nb) {
            int j = b / Tcur, t = b - j * Tcur;
            int k = j % Ns;
#pragma unroll
            for (int i = 0; i < R; ++i) x[u][i] = A[(j + i * m) * ld + t];
            if (Ns > 1) {
#pragma unroll
                for (int i = 1; i < R; ++i) x[u][i] = c_mulc(x[u][i], wL[i * k * twstep]);
            }
            Bfly<R>::run(x[u]);
            pos[u] = ((j - k) * R + k) * ld + t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        if (pos[u] >= 0) {
#pragma unroll
            for (int q = 0; q < R; ++q) A[pos[u] + q * Ns * ld] = x[u][q];
        }
    }
    __syncthreads();
}

template <bool VIN, bool VOUT>
__global__ void __launch_bounds__(256, 2)
pass_inplace_kernel(const PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fa_lds_raw[];
    cplx *A = reinterpret_cast<cplx *>(fa_lds_raw);
    const int tid = threadIdx.x, nth = blockDim.x;
    i64 blk = (i64)blockIdx.x + (i64)blockIdx.y * gridDim.x;
    i64 tile = blk % a.ntiles;
    i64 rest = blk / a.ntiles;
    i64 soff = 0, doff = 0, twb = 0;
    for (int d = 1; d < a.ndims; ++d) {
        i64 idx = rest % a.dn[d];
        rest /= a.dn[d];
        soff += idx * a.dis[d];
        doff += idx * a.dos[d];
        twb += idx * a.dtw[d];
    }
    const int Thi = a.T / a.lo_n;
    const i64 t0 = tile * Thi;
    const int Tcur = (int)((a.dn[0] - t0 < Thi) ? (a.dn[0] - t0) : Thi) * a.lo_n;
    const int L = a.L, ld = a.ld;
    const int total = L * Tcur;
    const bool tw_in = a.tw_n && (a.flags & FFTW_AMD_F_TW_IN);

    for (int e = tid; e < total; e += nth) {
        int l, t;
        if (a.in_t_fast) { l = e / Tcur; t = e - l * Tcur; }
        else             { t = e / L;    l = e - t * L; }
        const int thi = t / a.lo_n, tlo = t - thi * a.lo_n;
        i64 addr = soff + (i64)l * a.is_l + (t0 + thi) * a.dis[0] + tlo * a.lo_is;
        cplx v = load_elem<VIN>(a.src, addr, a.src_im, a.flags);
        if (tw_in) {
            i64 m = (i64)l * (twb + (t0 + thi) * a.dtw[0]);
            v = c_mulc(v, tw2(a.tw_lo, a.tw_hi, a.tw_shift, m));
        }
        A[l * ld + t] = v;
    }
    __syncthreads();

    int Ns = 1;
    for (int s = 0; s < a.nrad; ++s) {
        const int r = a.rad[s];
        switch (r) {
        case 2:  inplace_stage<2>(A, a.wL, L, ld, Tcur, Ns, tid); break;
        case 3:  inplace_stage<3>(A, a.wL, L, ld, Tcur, Ns, tid); break;
        case 4:  inplace_stage<4>(A, a.wL, L, ld, Tcur, Ns, tid); break;
        case 5:  inplace_stage<5>(A, a.wL, L, ld, Tcur, Ns, tid); break;
        case 7:  inplace_stage<7>(A, a.wL, L, ld, Tcur, Ns, tid); break;
        case 8:  inplace_stage<8>(A, a.wL, L, ld, Tcur, Ns, tid); break;
        case 11: inplace_stage<11>(A, a.wL, L, ld, Tcur, Ns, tid); break;
        case 13: inplace_stage<13>(A, a.wL, L, ld, Tcur, Ns, tid); break;
        default: inplace_stage<16>(A, a.wL, L, ld, Tcur, Ns, tid); break;
        }
        Ns *= r;
    }

    for (int e = tid; e < total; e += nth) {
        int l, t;
        if (a.out_t_fast) { l = e / Tcur; t = e - l * Tcur; }
        else              { t = e / L;    l = e - t * L; }
        cplx v = A[l * ld + t];
        const int thi = t / a.lo_n, tlo = t - thi * a.lo_n;
        if (a.tw_n && !tw_in) {
            i64 m = (i64)l * (twb + (t0 + thi) * a.dtw[0]);
            v = c_mulc(v, tw2(a.tw_lo, a.tw_hi, a.tw_shift, m));
        }
        i64 addr = doff + (i64)l * a.os_l + (t0 + thi) * a.dos[0] + tlo * a.lo_os;
        store_elem<VOUT>(a.dst, addr, a.dst_im, a.flags, v);
    }
}

/* ------------------------------------------------------------------------ */
/* strided copy / pad / multiply / permute                                   */
/* ------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------ */
/* index space of the element-wise kernels                                   */
/* ------------------------------------------------------------------------ */
/* Work items run over dims[0..kpos) (the loops that are more contiguous than the
   transform index on the user side), then the transform / pair index k in [0, K), then
   dims[kpos..ndims).  The part up to and including k is the "inner" index: one virtual
   block covers 256 consecutive inner indices of one combination of the outer dims, so
   the outer dims are peeled once per block and the inner ones with 32-bit arithmetic
   instead of a chain of 64-bit divisions per element.  (These kernels run at the
   device-to-device copy rate, 4.0-4.5 TB/s, either way: they are bound by their eight
   interleaved forward / mirrored streams, not by the index arithmetic.) */
struct ElemIdx {
    i64 dn[FFTW_AMD_MAX_DIMS], dis[FFTW_AMD_MAX_DIMS], dos[FFTW_AMD_MAX_DIMS];
    i64 nvb;                 /* virtual blocks = nblk * prod(dn[kpos..ndims)) */
    unsigned K, inner, nblk; /* inner = K * prod(dn[0..kpos)), nblk = ceil(inner / 256) */
    int ndims, kpos;
};

FA_DEV bool elem_index(const ElemIdx &e, i64 vb, i64 *k, i64 *soff, i64 *doff) {
    i64 ob = vb / e.nblk;
    unsigned i = (unsigned)(vb - ob * e.nblk) * 256u + threadIdx.x;
    i64 so = 0, dof = 0;
    for (int d = e.kpos; d < e.ndims; ++d) {       /* uniform over the block */
        i64 q = ob / e.dn[d], r = ob - q * e.dn[d];
        so += r * e.dis[d];
        dof += r * e.dos[d];
        ob = q;
    }
    if (i >= e.inner) return false;
    for (int d = 0; d < e.kpos; ++d) {
        unsigned n = (unsigned)e.dn[d], q = i / n, r = i - q * n;
        so += (i64)r * e.dis[d];
        dof += (i64)r * e.dos[d];
        i = q;
    }
    *k = i;
    *soff = so;
    *doff = dof;
    return true;
}

struct CopyArgs {
    const double *src;
    double *dst;
    i64 src_im, dst_im;
    i64 is_k, os_k;
    i64 K, Kvalid;
    ElemIdx e;
    const cplx *tab;
    const i64 *perm;
    int flags;
};

/* U virtual blocks (U x 256 consecutive inner indices) per trip of a workgroup: the U index chains (table entry ->
   element -> store) of a work-item are independent, so their loads overlap.  U = 4 for plain / padded / table
   copies (Bluestein's chirp products: 3.8 -> 3.0 ms per 2 GiB batch of n = 10007); the PERMUTED copies of Rader's
   gather / scatter stay at U = 1 -- they are bound by their 16-byte accesses to 128-byte lines (3.5 ms per 2 GiB
   where a pass takes 1.5) and four in flight per item change nothing (profiles/r03_prime_plan_steps.txt) */
template <int U>
__global__ void __launch_bounds__(256) copy_kernel(const CopyArgs a) {
    const i64 ngroups = (a.e.nvb + U - 1) / U;
    /* a permuted copy touches every 128-byte line of a row eight times, 16 bytes at a time: with the virtual blocks
       dealt round-robin all eight XCDs (workgroup b runs on XCD b % 8) fill / write back every line of every row.
       Keeping each ROW on one XCD leaves that to one L2: rows of 12288 points 3.5 -> 2.8 ms per 2 GiB, rows of 2^16
       points (1 MiB of the 4 MiB L2) unchanged */
    const i64 nrows = a.e.nvb / a.e.nblk;
    const bool by_rows = U == 1 && (gridDim.x & 7) == 0 && nrows >= 8;
    const i64 xcd = blockIdx.x & 7, per = by_rows ? (gridDim.x >> 3) : gridDim.x;
    for (i64 j = by_rows ? (blockIdx.x >> 3) : blockIdx.x; ; j += per) {
        i64 g = j;
        if (by_rows) {
            const i64 rr = j / a.e.nblk, row = xcd + 8 * rr;
            if (row >= nrows) break;
            g = row * a.e.nblk + (j - rr * a.e.nblk);
        } else if (g >= ngroups) break;
        i64 k[U], soff[U], doff[U], ks[U];
        bool ok[U];
        cplx v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const i64 vb = g * U + u;
            ok[u] = vb < a.e.nvb && elem_index(a.e, vb, &k[u], &soff[u], &doff[u]);
            if (!ok[u]) { k[u] = 0; soff[u] = 0; doff[u] = 0; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            ks[u] = (ok[u] && (a.flags & FFTW_AMD_F_PERM_SRC)) ? a.perm[k[u]] : k[u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = c_make(0.0, 0.0);
            if (ok[u] && k[u] < a.Kvalid) v[u] = load_elem<false>(a.src, soff[u] + ks[u] * a.is_k, a.src_im, a.flags);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            if (a.flags & FFTW_AMD_F_MUL_TABLE) v[u] = c_mul(v[u], a.tab[k[u]]);
            if (a.flags & FFTW_AMD_F_MUL_CONJ) v[u] = c_mulc(v[u], a.tab[k[u]]);
            const i64 kd = (a.flags & FFTW_AMD_F_PERM_DST) ? a.perm[k[u]] : k[u];
            store_elem<false>(a.dst, doff[u] + kd * a.os_k, a.dst_im, a.flags, v[u]);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* r2c untangle / c2r tangle                                                 */
/* ------------------------------------------------------------------------ */

struct RealArgs {
    const double *src;
    double *dst;
    i64 src_im, dst_im;
    i64 is_k, os_k;
    i64 h;       /* n / 2 */
    i64 npair;   /* h / 2 + 1 */
    ElemIdx e;
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
    int flags;
    int r2r;     /* fused r2r epilogue (r2c) / prologue (c2r): FFTW_AMD_R2R_* or 0 */
    int twmul;   /* untangle twiddle w_n^k = table entry k * twmul */
    i64 rn;      /* r2r length */
};

#include "r2r_epi.hpp"

/* Y[k] = E + w^k O, Y[h-k] = conj(E - w^k O), E = (Z[k] + conj Z[h-k]) / 2,
   O = -i (Z[k] - conj Z[h-k]) / 2   (the 1/2 is the
   KP500000000 of reference rdft_scalar/r2cf/hc2cfdft_4.c:137) */
__global__ void __launch_bounds__(256) r2c_post_kernel(const RealArgs a) {
    for (i64 vb = blockIdx.x; vb < a.e.nvb; vb += gridDim.x) {
        i64 k, soff, doff;
        if (!elem_index(a.e, vb, &k, &soff, &doff)) continue;
        i64 km = a.h - k;
        cplx zk = load_elem<false>(a.src, soff + k * a.is_k, a.src_im, 0);
        cplx zm = load_elem<false>(a.src, soff + (km == a.h ? 0 : km) * a.is_k, a.src_im, 0);
        cplx E = c_make(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
        cplx D = c_make(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));
        cplx O = c_mni(D);
        cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, k * a.twmul);
        cplx P = c_mulc(O, w);
        cplx yk = c_add(E, P);
        cplx ym = c_sub(E, P);
        ym.y = -ym.y;
        if (k == 0) { yk.y = 0.0; ym.y = 0.0; }
        epi_store(a, doff, k, yk);
        if (km != k) epi_store(a, doff, km, ym);
    }
}

/* Z'[k] = E' + i O', Z'[h-k] = conj(E' - i O'), E' = Y[k] + conj Y[h-k],
   O' = (Y[k] - conj Y[h-k]) w^-k  (transpose of the above; reference
   hc2cbdft codelets, no 1/2) */
__global__ void __launch_bounds__(256) c2r_pre_kernel(const RealArgs a) {
    for (i64 vb = blockIdx.x; vb < a.e.nvb; vb += gridDim.x) {
        i64 k, soff, doff;
        if (!elem_index(a.e, vb, &k, &soff, &doff)) continue;
        i64 km = a.h - k;
        cplx yk = pro_load(a, soff, k);
        cplx ym = pro_load(a, soff, km);
        if (k == 0) { yk.y = 0.0; ym.y = 0.0; }   /* Im Y[0], Im Y[n/2] are ignored */
        cplx E = c_make(yk.x + ym.x, yk.y - ym.y);
        cplx D = c_make(yk.x - ym.x, yk.y + ym.y);
        cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, k * a.twmul);
        cplx O = c_mul(D, w);
        cplx iO = c_mpi(O);
        cplx zk = c_add(E, iO);
        cplx zm = c_sub(E, iO);
        zm.y = -zm.y;
        store_elem<false>(a.dst, doff + k * a.os_k, a.dst_im, a.flags, zk);
        if (km != k && km != a.h)
            store_elem<false>(a.dst, doff + km * a.os_k, a.dst_im, a.flags, zm);
    }
}


/* The same two butterflies for the layout the large 1-D plans produce (interleaved complex on both sides,
   contiguous in k, the batch as the only loop, no r2r hook): 16-byte accesses, 32-bit index arithmetic, one
   work-item per pair (k, h-k); nontemporal on the caller's side.  Pure streaming. */
struct Real2Fast {
    const double *src;
    double *dst;
    i64 sbatch, dbatch;
    unsigned h, npair;
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
};
template <bool NT>
__global__ void __launch_bounds__(256) r2c_post_fast_kernel(const Real2Fast a) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= a.npair) return;
    const unsigned h = a.h, km = h - k;
    const double *s = a.src + (i64)blockIdx.y * a.sbatch;
    double *d = a.dst + (i64)blockIdx.y * a.dbatch;
    cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (i64)k);
    cplx zk = ld_cplx<false>(s + 2 * (i64)k);
    cplx zm = ld_cplx<false>(s + 2 * (i64)(km == h ? 0 : km));
    cplx E = c_make(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
    cplx D = c_make(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));
    cplx P = c_mulc(c_mni(D), w);
    cplx yk = c_add(E, P);
    cplx ym = c_sub(E, P);
    ym.y = -ym.y;
    if (k == 0) { yk.y = 0.0; ym.y = 0.0; }
    st_cplx<NT>(d + 2 * (i64)k, yk);
    if (km != k) st_cplx<NT>(d + 2 * (i64)km, ym);
}
template <bool NT>
__global__ void __launch_bounds__(256) c2r_pre_fast_kernel(const Real2Fast a) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= a.npair) return;
    const unsigned h = a.h, km = h - k;
    const double *s = a.src + (i64)blockIdx.y * a.sbatch;
    double *d = a.dst + (i64)blockIdx.y * a.dbatch;
    cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (i64)k);
    cplx yk = ld_cplx<NT>(s + 2 * (i64)k), ym = ld_cplx<NT>(s + 2 * (i64)km);
    if (k == 0) { yk.y = 0.0; ym.y = 0.0; }
    cplx E = c_make(yk.x + ym.x, yk.y - ym.y);
    cplx D = c_make(yk.x - ym.x, yk.y + ym.y);
    cplx iO = c_mpi(c_mul(D, w));
    cplx zk = c_add(E, iO);
    cplx zm = c_sub(E, iO);
    zm.y = -zm.y;
    st_cplx<false>(d + 2 * (i64)k, zk);
    if (km != k && km != h) st_cplx<false>(d + 2 * (i64)km, zm);
}

/* DCT-II / DST-II (REDFT10 / RODFT10) of long contiguous rows, streaming forms of the two element-wise steps
   (reference loops: reodft010e-r2hc, fftw/fftw_api.c:12465-12660):
   shuffle   v[j] = x[2j], v[n-1-j] = +-x[2j+1]: one work-item per input quad, 2 x 16 B in, 2 x 16 B out;
   untangle + epilogue: one work-item per pair (k, h-k) of the half-length spectrum Z (h = n / 2): Y[k], Y[h-k]
   as in r2c_post_fast_kernel, then y[k] = 2 Re(w^k Y[k]), y[n-k] = -2 Im(w^k Y[k]) with w = w_4n (the table's
   modulus; the untangle twiddle is its entry 4k, and w^(h-k) = e^(i pi/4) conj(w^k) saves a table lookup). */
struct DctFast {
    const double *src;
    double *dst;
    i64 sbatch, dbatch;
    unsigned n, nitems;
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
    int odd;            /* RODFT10: negate the odd samples / reverse the output */
};
template <bool NT>
__global__ void __launch_bounds__(256) dct2_shuffle_fast_kernel(const DctFast a) {
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= a.nitems) return;
    const double *s = a.src + (i64)blockIdx.y * a.sbatch + 4 * (i64)q;
    double *d = a.dst + (i64)blockIdx.y * a.dbatch;
    cplx u = ld_cplx<NT>(s), v = ld_cplx<NT>(s + 2);
    const double sg = a.odd ? -1.0 : 1.0;
    st_cplx<false>(d + 2 * (i64)q, c_make(u.x, v.x));
    st_cplx<false>(d + ((i64)a.n - 2 - 2 * (i64)q), c_make(sg * v.y, sg * u.y));
}
FA_DEV void dct2_epilogue(const DctFast &a, double *d, unsigned idx, cplx Y, cplx w) {
    const unsigned n = a.n;
    const bool mid = idx > 0 && 2 * idx < n;
    const double vi = mid ? Y.y : 0.0;
    const double dr = Y.x * w.x + vi * w.y, di = vi * w.x - Y.x * w.y;
    d[a.odd ? n - 1 - idx : idx] = 2.0 * dr;
    if (mid) d[a.odd ? idx - 1 : n - idx] = -2.0 * di;
}
template <bool NT>
__global__ void __launch_bounds__(256) dct2_untangle_fast_kernel(const DctFast a) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= a.nitems) return;
    const unsigned h = a.n / 2, km = h - k;
    const double *s = a.src + (i64)blockIdx.y * a.sbatch;
    double *d = a.dst + (i64)blockIdx.y * a.dbatch;
    cplx wu = tw2(a.tw_lo, a.tw_hi, a.tw_shift, 4 * (i64)k);      /* w_n^k */
    cplx wk = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (i64)k);          /* w_4n^k */
    cplx zk = ld_cplx<false>(s + 2 * (i64)k);
    cplx zm = ld_cplx<false>(s + 2 * (i64)(km == h ? 0 : km));
    cplx E = c_make(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
    cplx D = c_make(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));
    cplx P = c_mulc(c_mni(D), wu);
    cplx yk = c_add(E, P);
    cplx ym = c_sub(E, P);
    ym.y = -ym.y;
    if (k == 0) { yk.y = 0.0; ym.y = 0.0; }
    dct2_epilogue(a, d, k, yk, wk);
    if (km != k) {
        /* w_4n^(h-k) = w_4n^(n/2) conj(w_4n^k), w_4n^(n/2) = (cos, sin)(pi/4) */
        const cplx wm = c_make(FA_SQRT1_2 * (wk.x + wk.y), FA_SQRT1_2 * (wk.x - wk.y));
        dct2_epilogue(a, d, km, ym, wm);
    }
}

/* DCT-III / DST-III (REDFT01 / RODFT01) of long contiguous rows, the transposes of the two kernels above:
   prologue + tangle: Y[idx] = conj-twiddled (x[idx], x[n-idx]) (pro_load, r2r_epi.hpp), then the c2r tangle of
   the pair (k, h-k); unshuffle: y[2j] = v[j], y[2j+1] = +-v[n-1-j], one work-item per output quad. */
FA_DEV cplx dct3_prologue(const DctFast &a, const double *s, unsigned idx, cplx w) {
    const unsigned n = a.n;
    double x, y;
    if (!a.odd) { x = s[idx]; y = idx > 0 ? s[n - idx] : 0.0; }
    else { x = s[n - 1 - idx]; y = idx > 0 ? s[idx - 1] : 0.0; }
    return c_make(x * w.x + y * w.y, x * w.y - y * w.x);
}
template <bool NT>
__global__ void __launch_bounds__(256) dct3_tangle_fast_kernel(const DctFast a) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= a.nitems) return;
    const unsigned h = a.n / 2, km = h - k;
    const double *s = a.src + (i64)blockIdx.y * a.sbatch;
    double *d = a.dst + (i64)blockIdx.y * a.dbatch;
    cplx wu = tw2(a.tw_lo, a.tw_hi, a.tw_shift, 4 * (i64)k);      /* w_n^k */
    cplx wk = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (i64)k);          /* w_4n^k */
    const cplx wm = c_make(FA_SQRT1_2 * (wk.x + wk.y), FA_SQRT1_2 * (wk.x - wk.y));   /* w_4n^(h-k) */
    cplx yk = dct3_prologue(a, s, k, wk);
    cplx ym = dct3_prologue(a, s, km, wm);
    if (k == 0) { yk.y = 0.0; ym.y = 0.0; }
    cplx E = c_make(yk.x + ym.x, yk.y - ym.y);
    cplx D = c_make(yk.x - ym.x, yk.y + ym.y);
    cplx iO = c_mpi(c_mul(D, wu));
    cplx zk = c_add(E, iO);
    cplx zm = c_sub(E, iO);
    zm.y = -zm.y;
    st_cplx<false>(d + 2 * (i64)k, zk);
    if (km != k && km != h) st_cplx<false>(d + 2 * (i64)km, zm);
}
template <bool NT>
__global__ void __launch_bounds__(256) dct3_unshuffle_fast_kernel(const DctFast a) {
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= a.nitems) return;
    const double *s = a.src + (i64)blockIdx.y * a.sbatch;
    double *d = a.dst + (i64)blockIdx.y * a.dbatch + 4 * (i64)q;
    cplx lo = ld_cplx<false>(s + 2 * (i64)q);                          /* v[2q], v[2q+1] */
    cplx hi = ld_cplx<false>(s + ((i64)a.n - 2 - 2 * (i64)q));         /* v[n-2-2q], v[n-1-2q] */
    const double sg = a.odd ? -1.0 : 1.0;
    st_cplx<NT>(d, c_make(lo.x, sg * hi.y));
    st_cplx<NT>(d + 2, c_make(lo.y, sg * hi.x));
}

/* ------------------------------------------------------------------------ */
/* r2r pre / post processing                                                 */
/* ------------------------------------------------------------------------ */
/* One work item per index k of one transform; the flattened index runs over
   the loops in order of the user-side stride with k inserted at position kpos,
   so neighbouring work items touch neighbouring user elements whichever axis is
   transformed.  tw(m) = exp(+2 pi i m / M), M = 4n (kinds 01/10) or 8n (11).
   Index maps and their derivations: DESIGN.md section 9; the reference's loops
   with the same roles are cited at emit_r2r_axis in planner.c. */
struct R2RArgs {
    const double *src;
    double *dst;
    i64 src_im, dst_im;
    i64 is_k, os_k;
    i64 n, K;
    ElemIdx e;
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
    int mode;
};

__global__ void __launch_bounds__(256) r2r_kernel(const R2RArgs a) {
    const i64 n = a.n;
    for (i64 vb = blockIdx.x; vb < a.e.nvb; vb += gridDim.x) {
        i64 k, soff, doff;
        if (!elem_index(a.e, vb, &k, &soff, &doff)) continue;
        const double *S = a.src + soff;
        double *D = a.dst + doff;
#define SR(j) S[(j) * a.is_k]
#define SI(j) S[(j) * a.is_k + a.src_im]
#define DR(j) D[(j) * a.os_k]
#define DI(j) D[(j) * a.os_k + a.dst_im]
/* real sequences on the scratch side are addressed as pairs: element j at
   (j >> 1) * stride + (j & 1) * im  (planner.c emit_r2r_axis) */
#define DP(j) D[((j) >> 1) * a.os_k + ((j) & 1) * a.dst_im]
#define SP(j) S[((j) >> 1) * a.is_k + ((j) & 1) * a.src_im]
        switch (a.mode) {
        case FFTW_AMD_R2R_PRE_HC2R: {
            DR(k) = SR(k);
            DI(k) = (k > 0 && 2 * k < n) ? SR(n - k) : 0.0;
            break;
        }
        case FFTW_AMD_R2R_PRE_E10:
        case FFTW_AMD_R2R_PRE_O10: {
            /* v[j] = x[2j], v[n-1-j] = x[2j+1]: one work item per input pair */
            DP(k) = SR(2 * k);
            if (2 * k + 1 < n) {
                double b = SR(2 * k + 1);
                DP(n - 1 - k) = (a.mode == FFTW_AMD_R2R_PRE_O10) ? -b : b;
            }
            break;
        }
        case FFTW_AMD_R2R_PRE_E01:
        case FFTW_AMD_R2R_PRE_O01: {
            double x, y;
            if (a.mode == FFTW_AMD_R2R_PRE_E01) { x = SR(k); y = (k > 0) ? SR(n - k) : 0.0; }
            else { x = SR(n - 1 - k); y = (k > 0) ? SR(k - 1) : 0.0; }
            cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, k);
            DR(k) = x * w.x + y * w.y;
            DI(k) = x * w.y - y * w.x;
            break;
        }
        case FFTW_AMD_R2R_PRE_E00: {
            i64 N = 2 * (n - 1);
            DP(k) = SR(k < n ? k : N - k);
            break;
        }
        case FFTW_AMD_R2R_PRE_O00: {
            i64 N = 2 * (n + 1);
            double v = 0.0;
            if (k >= 1 && k <= n) v = SR(k - 1);
            else if (k > n + 1) v = -SR(N - k - 1);
            DP(k) = v;
            break;
        }
        case FFTW_AMD_R2R_PRE_E11:
        case FFTW_AMD_R2R_PRE_O11: {
            double xr = SR(2 * k), xi = SR(n - 1 - 2 * k);
            if (a.mode == FFTW_AMD_R2R_PRE_O11) { double t = xr; xr = xi; xi = t; }
            cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, 4 * k);
            DR(k) = xr * w.x + xi * w.y;
            DI(k) = xi * w.x - xr * w.y;
            break;
        }
        case FFTW_AMD_R2R_PRE_E11ODD:
        case FFTW_AMD_R2R_PRE_O11ODD: {
            double re = 0.0, im = 0.0;
            if (k < n) {
                double x = (a.mode == FFTW_AMD_R2R_PRE_E11ODD) ? SR(k) : SR(n - 1 - k);
                cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, 2 * k);
                re = x * w.x;
                im = -x * w.y;
            }
            DR(k) = re;
            DI(k) = im;
            break;
        }
        case FFTW_AMD_R2R_POST_R2HC: {
            DR(k) = SR(k);
            if (k > 0 && 2 * k < n) DR(n - k) = SI(k);
            break;
        }
        case FFTW_AMD_R2R_POST_DHT: {
            double re = SR(k);
            if (k > 0 && 2 * k < n) {
                double im = SI(k);
                DR(k) = re - im;
                DR(n - k) = re + im;
            } else {
                DR(k) = re;
            }
            break;
        }
        case FFTW_AMD_R2R_POST_E10:
        case FFTW_AMD_R2R_POST_O10: {
            const bool rev = (a.mode == FFTW_AMD_R2R_POST_O10);
            double vr = SR(k), vi = (k > 0 && 2 * k < n) ? SI(k) : 0.0;
            cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, k);
            double dr = vr * w.x + vi * w.y, di = vi * w.x - vr * w.y;
            DR(rev ? n - 1 - k : k) = 2.0 * dr;
            if (k > 0 && 2 * k < n) DR(rev ? k - 1 : n - k) = -2.0 * di;
            break;
        }
        case FFTW_AMD_R2R_POST_E01:
        case FFTW_AMD_R2R_POST_O01: {
            /* y[2j] = v[j], y[2j+1] = v[n-1-j]: one work item per output pair */
            DR(2 * k) = SP(k);
            if (2 * k + 1 < n) {
                double b = SP(n - 1 - k);
                DR(2 * k + 1) = (a.mode == FFTW_AMD_R2R_POST_O01) ? -b : b;
            }
            break;
        }
        case FFTW_AMD_R2R_POST_E00:
            DR(k) = SR(k);
            break;
        case FFTW_AMD_R2R_POST_O00:
            DR(k) = -SI(k + 1);
            break;
        case FFTW_AMD_R2R_POST_E11:
        case FFTW_AMD_R2R_POST_O11: {
            double zr = SR(k), zi = SI(k);
            cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, 4 * k + 1);
            double dr = zr * w.x + zi * w.y, di = zi * w.x - zr * w.y;
            DR(2 * k) = 2.0 * dr;
            DR(n - 1 - 2 * k) = (a.mode == FFTW_AMD_R2R_POST_O11) ? 2.0 * di : -2.0 * di;
            break;
        }
        case FFTW_AMD_R2R_POST_E11ODD:
        case FFTW_AMD_R2R_POST_O11ODD: {
            double zr = SR(k), zi = SI(k);
            cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, 2 * k + 1);
            double y = 2.0 * (zr * w.x + zi * w.y);
            if (a.mode == FFTW_AMD_R2R_POST_O11ODD && (k & 1)) y = -y;
            DR(k) = y;
            break;
        }
        default:
            break;
        }
#undef SR
#undef SI
#undef DR
#undef DI
#undef DP
#undef SP
    }
}

/* ------------------------------------------------------------------------ */
/* radix-4 r2c untangle / c2r tangle                                         */
/* ------------------------------------------------------------------------ */
/* n = 4m.  z_v[j] = x[4j+2v] + i x[4j+2v+1] (v = 0,1), Z_v = DFT_m(z_v) stored
   as [v][m].  With X_s = DFT_m(x[4j+s]):  X_{2v} = (Z_v[k] + conj Z_v[m-k]) / 2,
   X_{2v+1} = -i (Z_v[k] - conj Z_v[m-k]) / 2, T_s = w_n^(sk) X_s, and
       Y[k]    = T0 + T1 + T2 + T3        Y[k+m]  = T0 - iT1 - T2 + iT3
       Y[2m-k] = conj(T0 - T1 + T2 - T3)  Y[m-k]  = conj(T0 + iT1 - T2 - iT3)
   This is the reference's rdft2-ct-dit/4 step with the hc2cfdft_4 codelet
   (fftw/fftw_api.c:5579-5590, fftw/rdft_scalar/r2cf/hc2cfdft_4.c:135-212), the
   plan it picks for n = 2^22. */
struct Real4Args {
    const double *src;   /* Z: element k of vector v at src + v*vs + k*is_k */
    double *dst;
    i64 src_im, dst_im;
    i64 is_k, vs, os_k;
    i64 m, npair;
    ElemIdx e;
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
    int flags;
    int r2r, twmul;      /* as in RealArgs */
    i64 rn;
};

__global__ void __launch_bounds__(256) r2c_post4_kernel(const Real4Args a) {
    for (i64 vb = blockIdx.x; vb < a.e.nvb; vb += gridDim.x) {
        i64 k, soff, doff;
        if (!elem_index(a.e, vb, &k, &soff, &doff)) continue;
        const i64 m = a.m, km = (k == 0) ? 0 : m - k;
        cplx T[4];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            cplx zk = load_elem<false>(a.src, soff + v * a.vs + k * a.is_k, a.src_im, 0);
            cplx zm = load_elem<false>(a.src, soff + v * a.vs + km * a.is_k, a.src_im, 0);
            cplx E = c_make(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
            cplx D = c_make(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));
            T[2 * v] = E;
            T[2 * v + 1] = c_mni(D);
        }
        T[1] = c_mulc(T[1], tw2(a.tw_lo, a.tw_hi, a.tw_shift, k * a.twmul));
        T[2] = c_mulc(T[2], tw2(a.tw_lo, a.tw_hi, a.tw_shift, 2 * k * a.twmul));
        T[3] = c_mulc(T[3], tw2(a.tw_lo, a.tw_hi, a.tw_shift, 3 * k * a.twmul));
        cplx s02 = c_add(T[0], T[2]), d02 = c_sub(T[0], T[2]);
        cplx s13 = c_add(T[1], T[3]), d13 = c_sub(T[1], T[3]);
        cplx y0 = c_add(s02, s13);                 /* Y[k]       */
        cplx y1 = c_add(d02, c_mni(d13));          /* Y[k+m]     */
        cplx y2 = c_sub(s02, s13);  y2.y = -y2.y;  /* Y[2m-k]    */
        cplx y3 = c_add(d02, c_mpi(d13)); y3.y = -y3.y;   /* Y[m-k] */
        if (k == 0) { y0.y = 0.0; y2.y = 0.0; }
        epi_store(a, doff, k, y0);
        epi_store(a, doff, k + m, y1);
        epi_store(a, doff, 2 * m - k, y2);
        if (k != 0 && 2 * k != m) epi_store(a, doff, m - k, y3);
    }
}

/* The same butterfly for the layout the large 1-D plans produce -- Z_0[k], Z_1[k] side by side
   (32 contiguous bytes per k), interleaved complex output, no r2r epilogue, one loop dim (the
   batch): 16-byte accesses, 32-bit index arithmetic, one work-item per pair (k, m-k), w^2k and
   w^3k from w^k instead of four more table loads.  Pure streaming: 4 x 16 B in, 4 x 16 B out. */
struct Real4Fast {
    const double *src;
    double *dst;
    i64 sbatch, dbatch;   /* distance between transforms, in doubles */
    unsigned m, npair;
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
    int nt_out;
};

template <bool NT>
__global__ void __launch_bounds__(256) r2c_post4_fast_kernel(const Real4Fast a) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= a.npair) return;
    const unsigned m = a.m, km = k ? m - k : 0;
    const double *s = a.src + (i64)blockIdx.y * a.sbatch;
    double *d = a.dst + (i64)blockIdx.y * a.dbatch;
    cplx w1 = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (i64)k);
    cplx zk0 = ld_cplx<false>(s + 4 * (i64)k), zk1 = ld_cplx<false>(s + 4 * (i64)k + 2);
    cplx zm0 = ld_cplx<false>(s + 4 * (i64)km), zm1 = ld_cplx<false>(s + 4 * (i64)km + 2);
    cplx T0 = c_make(0.5 * (zk0.x + zm0.x), 0.5 * (zk0.y - zm0.y));
    cplx T1 = c_mni(c_make(0.5 * (zk0.x - zm0.x), 0.5 * (zk0.y + zm0.y)));
    cplx T2 = c_make(0.5 * (zk1.x + zm1.x), 0.5 * (zk1.y - zm1.y));
    cplx T3 = c_mni(c_make(0.5 * (zk1.x - zm1.x), 0.5 * (zk1.y + zm1.y)));
    cplx w2 = c_mul(w1, w1), w3 = c_mul(w2, w1);
    T1 = c_mulc(T1, w1);
    T2 = c_mulc(T2, w2);
    T3 = c_mulc(T3, w3);
    cplx s02 = c_add(T0, T2), d02 = c_sub(T0, T2);
    cplx s13 = c_add(T1, T3), d13 = c_sub(T1, T3);
    cplx y0 = c_add(s02, s13);                 /* Y[k]       */
    cplx y1 = c_add(d02, c_mni(d13));          /* Y[k+m]     */
    cplx y2 = c_sub(s02, s13);  y2.y = -y2.y;  /* Y[2m-k]    */
    cplx y3 = c_add(d02, c_mpi(d13)); y3.y = -y3.y;   /* Y[m-k] */
    if (k == 0) { y0.y = 0.0; y2.y = 0.0; }
    st_cplx<NT>(d + 2 * (i64)k, y0);
    st_cplx<NT>(d + 2 * ((i64)k + m), y1);
    st_cplx<NT>(d + 2 * (2 * (i64)m - k), y2);
    if (k != 0 && 2 * k != m) st_cplx<NT>(d + 2 * ((i64)m - k), y3);
}

/* c2r: with A = Y[k], B = Y[k+m], C = conj Y[2m-k], D = conj Y[m-k]:
       S0 = A+B+C+D, S1 = A+iB-C-iD, S2 = A-B+C-D, S3 = A-iB-C+iD,
       X'_s = w_n^(-sk) S_s,  Z'_0[k] = X'_0 + i X'_1,  Z'_1[k] = X'_2 + i X'_3;
   the mirror index m-k uses the same four loads with conjugated roles. */
FA_DEV void c2r4_combine(cplx A, cplx B, cplx C, cplx D, cplx w1, cplx w2, cplx w3, cplx *z0, cplx *z1) {
    cplx sAC = c_add(A, C), dAC = c_sub(A, C), sBD = c_add(B, D), dBD = c_sub(B, D);
    cplx S0 = c_add(sAC, sBD);
    cplx S1 = c_add(dAC, c_mpi(dBD));
    cplx S2 = c_sub(sAC, sBD);
    cplx S3 = c_add(dAC, c_mni(dBD));
    cplx X1 = c_mul(S1, w1), X2 = c_mul(S2, w2), X3 = c_mul(S3, w3);
    *z0 = c_add(S0, c_mpi(X1));
    *z1 = c_add(X2, c_mpi(X3));
}

__global__ void __launch_bounds__(256) c2r_pre4_kernel(const Real4Args a) {
    for (i64 vb = blockIdx.x; vb < a.e.nvb; vb += gridDim.x) {
        i64 k, soff, doff;
        if (!elem_index(a.e, vb, &k, &soff, &doff)) continue;
        const i64 m = a.m;
        /* the four half-spectrum entries this pair needs (src here is Y, is_k its stride) */
        cplx Yk = pro_load(a, soff, k);
        cplx Ykm = pro_load(a, soff, k + m);
        cplx Y2 = pro_load(a, soff, 2 * m - k);
        cplx Y1 = pro_load(a, soff, m - k);
        if (k == 0) { Yk.y = 0.0; Y2.y = 0.0; }      /* Im Y[0], Im Y[n/2] are ignored */
        cplx w1 = tw2(a.tw_lo, a.tw_hi, a.tw_shift, k * a.twmul);
        cplx w2 = tw2(a.tw_lo, a.tw_hi, a.tw_shift, 2 * k * a.twmul);
        cplx w3 = tw2(a.tw_lo, a.tw_hi, a.tw_shift, 3 * k * a.twmul);
        cplx z0, z1;
        c2r4_combine(Yk, Ykm, c_make(Y2.x, -Y2.y), c_make(Y1.x, -Y1.y), w1, w2, w3, &z0, &z1);
        store_elem<false>(a.dst, doff + k * a.os_k, a.dst_im, a.flags, z0);
        store_elem<false>(a.dst, doff + a.vs + k * a.os_k, a.dst_im, a.flags, z1);
        if (k != 0 && 2 * k != m) {
            /* mirror k' = m-k: A' = Y[m-k], B' = Y[2m-k], C' = conj Y[m+k], D' = conj Y[k];
               w_n^(-s(m-k)) = i^s conj(w_s) */
            cplx v1 = c_mpi(c_make(w1.x, -w1.y));
            cplx v2 = c_make(-w2.x, w2.y);
            cplx v3 = c_mni(c_make(w3.x, -w3.y));
            c2r4_combine(Y1, Y2, c_make(Ykm.x, -Ykm.y), c_make(Yk.x, -Yk.y), v1, v2, v3, &z0, &z1);
            store_elem<false>(a.dst, doff + (m - k) * a.os_k, a.dst_im, a.flags, z0);
            store_elem<false>(a.dst, doff + a.vs + (m - k) * a.os_k, a.dst_im, a.flags, z1);
        }
    }
}

/* the transpose of r2c_post4_fast_kernel: interleaved half spectrum in, Z_0[k], Z_1[k] side by side out */
template <bool NT>
__global__ void __launch_bounds__(256) c2r_pre4_fast_kernel(const Real4Fast a) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= a.npair) return;
    const unsigned m = a.m;
    const double *s = a.src + (i64)blockIdx.y * a.sbatch;
    double *d = a.dst + (i64)blockIdx.y * a.dbatch;
    cplx w1 = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (i64)k);
    cplx Yk = ld_cplx<NT>(s + 2 * (i64)k), Ykm = ld_cplx<NT>(s + 2 * ((i64)k + m));
    cplx Y2 = ld_cplx<NT>(s + 2 * (2 * (i64)m - k)), Y1 = ld_cplx<NT>(s + 2 * ((i64)m - k));
    if (k == 0) { Yk.y = 0.0; Y2.y = 0.0; }
    cplx w2 = c_mul(w1, w1), w3 = c_mul(w2, w1);
    cplx z0, z1;
    c2r4_combine(Yk, Ykm, c_make(Y2.x, -Y2.y), c_make(Y1.x, -Y1.y), w1, w2, w3, &z0, &z1);
    st_cplx<false>(d + 4 * (i64)k, z0);
    st_cplx<false>(d + 4 * (i64)k + 2, z1);
    if (k != 0 && 2 * k != m) {
        cplx v1 = c_mpi(c_make(w1.x, -w1.y));
        cplx v2 = c_make(-w2.x, w2.y);
        cplx v3 = c_mni(c_make(w3.x, -w3.y));
        c2r4_combine(Y1, Y2, c_make(Ykm.x, -Ykm.y), c_make(Yk.x, -Yk.y), v1, v2, v3, &z0, &z1);
        st_cplx<false>(d + 4 * ((i64)m - k), z0);
        st_cplx<false>(d + 4 * ((i64)m - k) + 2, z1);
    }
}

/* Rader: P[k] = A[k] * Omega[k]; P[0] += x0; Y[0] = x0 + A[0]
   (reference rader_apply A.c:4218-4240) */
struct RaderArgs {
    double *work;        /* [vec][p-1] complex, contiguous */
    const double *x0;    /* [vec] complex */
    double *dst;         /* where Y[0] goes */
    i64 dst_im;
    i64 pm1, nvec, total;
    i64 dn[FFTW_AMD_MAX_DIMS], dos[FFTW_AMD_MAX_DIMS];
    const cplx *omega;
    int ndims, flags;
};

__global__ void __launch_bounds__(256) rader_mul_kernel(const RaderArgs a) {
    i64 gid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 stride = (i64)gridDim.x * blockDim.x;
    for (; gid < a.total; gid += stride) {
        i64 k = gid % a.pm1;
        i64 v = gid / a.pm1;
        cplx *w = reinterpret_cast<cplx *>(a.work) + v * a.pm1;
        cplx A = w[k];
        cplx P = c_mul(A, a.omega[k]);
        if (k == 0) {
            cplx x0 = reinterpret_cast<const cplx *>(a.x0)[v];
            P = c_add(P, x0);
            i64 rest = v, doff = 0;
            for (int d = 0; d < a.ndims; ++d) {
                i64 idx = rest % a.dn[d];
                rest /= a.dn[d];
                doff += idx * a.dos[d];
            }
            store_elem<false>(a.dst, doff, a.dst_im, a.flags, c_add(x0, A));
        }
        w[k] = P;
    }
}

/* half spectrum Y[0..n/2] -> full Hermitian spectrum F[0..n-1] (odd-n c2r) */
__global__ void __launch_bounds__(256) herm_expand_kernel(const CopyArgs a) {
    for (i64 vb = blockIdx.x; vb < a.e.nvb; vb += gridDim.x) {
        i64 k, soff, doff;
        if (!elem_index(a.e, vb, &k, &soff, &doff)) continue;
        i64 half = a.K / 2;
        i64 ks = (k <= half) ? k : a.K - k;
        cplx v = load_elem<false>(a.src, soff + ks * a.is_k, a.src_im, 0);
        if (k > half) v.y = -v.y;
        if (k == 0 || (2 * k == a.K)) v.y = 0.0;
        store_elem<false>(a.dst, doff + k * a.os_k, a.dst_im, a.flags, v);
    }
}

/* ------------------------------------------------------------------------ */
/* host side                                                                 */
/* ------------------------------------------------------------------------ */

static int g_dev_count = -1;

extern "C" int fa_hip_device_count(void) {
    if (g_dev_count < 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
        g_dev_count = n;
    }
    return g_dev_count;
}

/* NULL when the device cannot provide the memory (the planners then return NULL, like the reference's
   planner does when a solver's buffers cannot be had); every other HIP error is fatal */
extern "C" void *fa_hip_malloc(size_t nbytes) {
    void *p = NULL;
    if (nbytes == 0) nbytes = 16;
    hipError_t e = hipMalloc(&p, nbytes);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
        (void)hipGetLastError();
        fprintf(stderr, "fftw3_amd: the device cannot allocate %zu bytes\n", nbytes);
        return NULL;
    }
    FA_CHECK(e);
    return p;
}

extern "C" void fa_hip_free(void *p) {
    if (p) FA_CHECK(hipFree(p));
}

extern "C" void *fa_hip_host_malloc(size_t nbytes) {
    if (fa_hip_device_count() <= 0) return NULL;
    void *p = NULL;
    if (hipHostMalloc(&p, nbytes ? nbytes : 16, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return NULL;
    }
    return p;
}

extern "C" int fa_hip_host_free(void *p) {
    if (fa_hip_device_count() <= 0) return 0;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (attr.type == hipMemoryTypeHost) { FA_CHECK(hipHostFree(p)); return 1; }
    return 0;
}

extern "C" int fa_hip_is_device_ptr(const void *p) {
    if (!p || fa_hip_device_count() <= 0) return 0;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

/* ordinal of the device that owns a device allocation, -1 for anything else */
extern "C" int fa_hip_ptr_device(const void *p) {
    if (!p || fa_hip_device_count() <= 0) return -1;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged) return -1;
    return attr.device;
}

extern "C" void fa_hip_memcpy_h2d(void *dst, const void *src, size_t n, void *stream) {
    FA_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, (hipStream_t)stream));
}
extern "C" void fa_hip_memcpy_d2h(void *dst, const void *src, size_t n, void *stream) {
    FA_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, (hipStream_t)stream));
}
extern "C" void fa_hip_memset(void *dst, int v, size_t n, void *stream) {
    FA_CHECK(hipMemsetAsync(dst, v, n, (hipStream_t)stream));
}
extern "C" void fa_hip_stream_sync(void *stream) {
    FA_CHECK(hipStreamSynchronize((hipStream_t)stream));
}

extern "C" void *fa_hip_event_create(void) {
    hipEvent_t e;
    FA_CHECK(hipEventCreate(&e));
    return (void *)e;
}
extern "C" void fa_hip_event_record(void *ev, void *stream) {
    FA_CHECK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
}
extern "C" float fa_hip_event_elapsed_ms(void *a, void *b) {
    float ms = 0.f;
    FA_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b));
    return ms;
}
extern "C" void fa_hip_event_destroy(void *ev) { FA_CHECK(hipEventDestroy((hipEvent_t)ev)); }
extern "C" void *fa_hip_stream_create(void) {
    hipStream_t s;
    FA_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return (void *)s;
}
extern "C" void fa_hip_stream_destroy(void *s) { FA_CHECK(hipStreamDestroy((hipStream_t)s)); }
extern "C" int fa_hip_get_device(void) { int d = 0; FA_CHECK(hipGetDevice(&d)); return d; }
extern "C" void fa_hip_set_device(int dev) { FA_CHECK(hipSetDevice(dev)); }
/* 0 on success; a missing peer path is not an error (hipMemcpyPeerAsync stages through the host then) */
extern "C" int fa_hip_enable_peer(int dev, int peer) {
    int can = 0, cur = 0;
    if (dev == peer) return 0;
    FA_CHECK(hipGetDevice(&cur));
    if (hipDeviceCanAccessPeer(&can, dev, peer) != hipSuccess || !can) { (void)hipGetLastError(); return 1; }
    FA_CHECK(hipSetDevice(dev));
    hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
    if (e != hipSuccess) (void)hipGetLastError();            /* hipErrorPeerAccessAlreadyEnabled included */
    FA_CHECK(hipSetDevice(cur));
    return (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ? 0 : 1;
}
extern "C" void fa_hip_memcpy_peer(void *dst, int dst_dev, const void *src, int src_dev, size_t n, void *stream) {
    if (dst_dev == src_dev) FA_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    else FA_CHECK(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, n, (hipStream_t)stream));
}
extern "C" void fa_hip_memcpy2d_peer(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, void *stream) {
    if (!width || !height) return;
    FA_CHECK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, (hipStream_t)stream));
}
extern "C" void fa_hip_stream_wait_event(void *s, void *ev) {
    FA_CHECK(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)ev, 0));
}


static void grid_for(i64 total, dim3 *grid) {
    i64 blocks = (total + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 32) blocks = 256 * 32;   /* grid-stride beyond that */
    *grid = dim3((unsigned)blocks, 1, 1);
}


static std::atomic<unsigned> g_lds_attr_done{0};

template <bool VIN, bool VOUT>
static void launch_pass_variant(const PassArgs &pa, dim3 grid, size_t lds, hipStream_t st) {
    static int nth = 0;
    if (!nth) { const char *e = getenv("FFTW_AMD_GENERIC_THREADS"); nth = e ? atoi(e) : 256; if (nth < 64 || nth > 256) nth = 256; }
    hipLaunchKernelGGL((pass_generic_kernel<VIN, VOUT>), grid, dim3(nth), lds, st, pa);
}

/* kernels_rr.hip */
int fa_launch_pass3tw(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                      i64 cs, i64 cn, hipStream_t st);          /* kernels_r3tw.hip */
int fa_launch_pass3gw(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                      i64 cs, i64 cn, hipStream_t st);          /* kernels_r3w.hip */
int fa_launch_pass3g(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                     i64 cs, i64 cn, hipStream_t st);
int fa_launch_pass3t(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                     i64 cs, i64 cn, hipStream_t st);
int fa_launch_r2crows(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                      i64 cs, i64 cn, hipStream_t st);
int fa_launch_pass3s(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                     i64 cs, i64 cn, hipStream_t st);
int fa_launch_passrr(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                     i64 cs, i64 cn, hipStream_t st);
/* kernels_blue.hip */
int fa_launch_blue(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                   i64 cs, i64 cn, hipStream_t st);
/* kernels_r1.hip */
int fa_launch_pass1r(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                     i64 cs, i64 cn, hipStream_t st);

template <bool IN_T, bool OUT_T, int HAS_TW>
static void launch_p1024_variant(const P1024Args &pa, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    const size_t lds = FA_P1024_LDS_DOUBLES * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass1024_kernel<IN_T, OUT_T, HAS_TW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL((pass1024_kernel<IN_T, OUT_T, HAS_TW>), grid, dim3(256), lds, st, pa);
}

/* arguments of the register-resident 1024-point pass for one step and chunk; returns 1 if the step
   does not qualify (caller falls through to the generic kernel), 0 and *nblocks_out == 0 for an empty launch */
static int fill_p1024(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                      i64 cs, i64 cn, P1024Args &pa, i64 *nblocks_out, bool *in_t, bool *out_t, int *tw) {
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    if (d->L != 1024 || d->src_im != 1 || d->dst_im != 1 ||
        (d->flags & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT | FFTW_AMD_F_CONJ_OUT)))
        return 1;
    for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) {
        pa.dn[i] = (i < d->ndims) ? d->dim_n[i] : 1;
        pa.dis[i] = (i < d->ndims) ? d->dim_is[i] : 0;
        pa.dos[i] = (i < d->ndims) ? d->dim_os[i] : 0;
        pa.dtw[i] = (i < d->ndims) ? d->dim_tw[i] : 0;
    }
    if (bd >= 0) {
        sbase += chunk_adv(d->src_buf, cs, d->dim_is[bd]);
        dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
        pa.dn[bd] = cn;
    }
    pa.src = bufs[d->src_buf] + sbase;
    pa.dst = bufs[d->dst_buf] + dbase;
    pa.is_l = d->is_l;
    pa.os_l = d->os_l;
    if (((uintptr_t)pa.src % 16) || ((uintptr_t)pa.dst % 16) || (pa.is_l % 2) || (pa.os_l % 2)) return 1;
    for (int i = 0; i < d->ndims; ++i)
        if ((pa.dis[i] % 2) || (pa.dos[i] % 2)) return 1;
    pa.w1024 = (const cplx *)tables[d->table];
    pa.tw_shift = d->tw_shift;
    pa.tw_lo = d->tw_n ? (const cplx *)tables[d->tw_lo] : NULL;
    pa.tw_hi = d->tw_n ? (const cplx *)tables[d->tw_hi] : NULL;
    pa.ndims = d->ndims;
    pa.flags = d->flags;
    pa.lo_sh = 0; pa.lo_is = d->tile_lo_is; pa.lo_os = d->tile_lo_os;
    pa.dbg = NULL;
    if (d->tile_lo_n > 1) {
        if (d->tile_lo_n != 2 && d->tile_lo_n != 4) return 1;
        pa.lo_sh = d->tile_lo_n == 2 ? 1 : 2;
        if ((pa.lo_is % 2) || (pa.lo_os % 2)) return 1;
    }
    pa.ntiles = (pa.dn[0] + (8 >> pa.lo_sh) - 1) / (8 >> pa.lo_sh);
    i64 nblocks = pa.ntiles;
    for (int i = 1; i < d->ndims; ++i) nblocks *= pa.dn[i];
    if (nblocks > 0x7fffffffLL) return 1;
    *nblocks_out = nblocks < 0 ? 0 : nblocks;
    *in_t = pa.dn[0] > 1 && iabs64(pa.dis[0]) <= iabs64(pa.is_l);
    *out_t = pa.dn[0] > 1 && iabs64(pa.dos[0]) <= iabs64(pa.os_l);
    *tw = d->tw_n == 0 ? 0 : ((d->flags & FFTW_AMD_F_TW_IN) ? 2 : 1);
    return 0;
}

static int launch_p1024(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                        i64 cs, i64 cn, hipStream_t st) {
    P1024Args pa;
    i64 nblocks = 0;
    bool in_t, out_t;
    int tw;
    if (fill_p1024(d, bufs, tables, cs, cn, pa, &nblocks, &in_t, &out_t, &tw)) return 1;
    if (nblocks <= 0) return 0;
    dim3 grid((unsigned)nblocks, 1, 1);
#define FA_P1024_CASE(I, O, W) if (in_t == I && out_t == O && tw == W) { launch_p1024_variant<I, O, W>(pa, grid, st); return 0; }
    FA_P1024_CASE(true, true, 0)  FA_P1024_CASE(true, true, 1)  FA_P1024_CASE(true, true, 2)
    FA_P1024_CASE(false, true, 0) FA_P1024_CASE(false, true, 1) FA_P1024_CASE(false, true, 2)
    FA_P1024_CASE(true, false, 0) FA_P1024_CASE(true, false, 1) FA_P1024_CASE(true, false, 2)
    FA_P1024_CASE(false, false, 0) FA_P1024_CASE(false, false, 1) FA_P1024_CASE(false, false, 2)
#undef FA_P1024_CASE
    return 1;
}

/* one tile of a launch described by `a`, block id `b0` of `nb` (XCD-contiguous order) */
template <bool IN_T, bool OUT_T, int HAS_TW>
FA_DEV void p1024_block(const P1024Args &a, unsigned b0, unsigned nb, double *plane) {
    unsigned blk = (nb & 7) ? b0 : (b0 & 7) * (nb >> 3) + (b0 >> 3);
    const unsigned nt = (unsigned)a.ntiles;
    unsigned rest = blk / nt;
    const unsigned tile = blk - rest * nt;
    i64 soff = 0, doff = 0, twb = 0;
    for (int d = 1; d < a.ndims; ++d) {
        const unsigned dn = (unsigned)a.dn[d];
        const unsigned q = rest / dn, idx = rest - q * dn;
        rest = q;
        soff += (i64)idx * a.dis[d];
        doff += (i64)idx * a.dos[d];
        twb += (i64)idx * a.dtw[d];
    }
    const i64 t0 = (i64)tile * 8;
    P1024Tile t;
    t.lo_sh = 0; t.lo_is = 0; t.lo_os = 0;
    t.src = a.src + soff + t0 * a.dis[0];
    t.dst = a.dst + doff + t0 * a.dos[0];
    t.is_l = a.is_l; t.os_l = a.os_l;
    t.dis0 = a.dis[0]; t.dos0 = a.dos[0];
    t.dtw0 = a.dtw[0]; t.q0 = twb + t0 * a.dtw[0];
    t.w1024 = a.w1024; t.tw_lo = a.tw_lo; t.tw_hi = a.tw_hi; t.tw_shift = a.tw_shift;
    t.Tcur = (int)((a.dn[0] - t0 < 8) ? (a.dn[0] - t0) : 8);
    t.flags = a.flags;
    t.dbg = NULL;
    p1024_tile<IN_T, OUT_T, HAS_TW, 0>(t, plane, threadIdx.x);
}

/* blocks [0, n2): row pass with input twiddle (pass 2 of the previous chunk); blocks [n2, n2 + n1):
   column pass (pass 1 of this chunk).  See fa_hip_launch_pair1024. */
__global__ void __launch_bounds__(256, 2)
pass1024_pair_kernel(const P1024Args a2, const P1024Args a1, const unsigned n2, const unsigned n1) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    if (blockIdx.x < n2) p1024_block<false, true, 2>(a2, blockIdx.x, n2, plane);
    else p1024_block<true, true, 0>(a1, blockIdx.x - n2, n1, plane);
}

/* The two passes of the batched N = 1024 x 1024 plan in ONE launch per chunk: first the tiles of
   pass 2 of the previous chunk, then the tiles of pass 1 of this chunk (pass1024.hpp,
   pass1024_pair_kernel).  Workgroups are dispatched in order, so pass 1 of chunk c fills the slots
   that the tail of pass 2 of chunk c-1 leaves idle: one dependent launch boundary per chunk instead
   of two.  Either half may be empty (first / last launch).  Returns 1 when the steps are not the
   (column pass without twiddle, row pass with input twiddle) pair the kernel is built for. */
extern "C" int fa_hip_launch_pair1024(const fftw_amd_step_desc *d_second, double *const *bufs_second,
                                      long long cs2, long long cn2,
                                      const fftw_amd_step_desc *d_first, double *const *bufs_first,
                                      long long cs1, long long cn1, void *const *tables, void *stream) {
    P1024Args a2, a1;
    i64 n2 = 0, n1 = 0;
    bool i2 = false, o2 = true, i1 = true, o1 = true;
    int w2 = 2, w1 = 0;
    memset((void *)&a2, 0, sizeof(a2));
    memset((void *)&a1, 0, sizeof(a1));
    if (cn2 > 0 && fill_p1024(d_second, bufs_second, tables, cs2, cn2, a2, &n2, &i2, &o2, &w2)) return 1;
    if (cn1 > 0 && fill_p1024(d_first, bufs_first, tables, cs1, cn1, a1, &n1, &i1, &o1, &w1)) return 1;
    if ((cn2 > 0 && (i2 || !o2 || w2 != 2 || a2.lo_sh)) || (cn1 > 0 && (!i1 || !o1 || w1 != 0 || a1.lo_sh))) return 1;
    if (n1 + n2 <= 0) return 0;
    if (n1 + n2 > 0x7fffffffLL) return 1;
    static std::atomic<unsigned> attr_done{0};
    const size_t lds = FA_P1024_LDS_DOUBLES * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass1024_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL(pass1024_pair_kernel, dim3((unsigned)(n1 + n2)), dim3(256), lds, (hipStream_t)stream,
                       a2, a1, (unsigned)n2, (unsigned)n1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fprintf(stderr, "fftw3_amd: pair launch failed: %s\n", hipGetErrorString(e)); return -1; }
    return 0;
}

static int launch_pass(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                       i64 cs, i64 cn, hipStream_t st) {
    PassArgs pa;
    if (d->flags & (FFTW_AMD_F_R2C_ROWS | FFTW_AMD_F_C2R_ROWS)) return fa_launch_r2crows(d, bufs, tables, cs, cn, st);
    if (d->flags & FFTW_AMD_F_REAL_DEC) {
        /* the last trip of a two-trip r2c: one executor (kernels_r3tw.hip), planned only where it applies */
        if (fa_launch_pass3tw(d, bufs, tables, cs, cn, st) == 0) return 0;
        fprintf(stderr, "fftw3_amd: internal error: real-decimated rows step with an unsupported layout\n");
        abort();
    }
    if (d->variant == FFTW_AMD_K_P1024 && launch_p1024(d, bufs, tables, cs, cn, st) == 0) return 0;
    if (d->variant == FFTW_AMD_K_BLUE) return fa_launch_blue(d, bufs, tables, cs, cn, st);
    if (d->variant == FFTW_AMD_K_R1 && fa_launch_pass1r(d, bufs, tables, cs, cn, st) == 0) return 0;
    if (d->variant == FFTW_AMD_K_RR && fa_launch_passrr(d, bufs, tables, cs, cn, st) == 0) return 0;
    if (d->variant == FFTW_AMD_K_R3 && d->L > 8192 && d->L < 16384 && fa_launch_pass3gw(d, bufs, tables, cs, cn, st) == 0) return 0;
    if (d->variant == FFTW_AMD_K_R3 && d->L > 1024 && d->L <= 2048 && fa_launch_pass3tw(d, bufs, tables, cs, cn, st) == 0) return 0;
    if (d->variant == FFTW_AMD_K_R3 && (fa_launch_pass3s(d, bufs, tables, cs, cn, st) == 0 ||
                                        fa_launch_pass3g(d, bufs, tables, cs, cn, st) == 0 ||
                                        fa_launch_pass3t(d, bufs, tables, cs, cn, st) == 0)) return 0;
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) {
        pa.dn[i] = (i < d->ndims) ? d->dim_n[i] : 1;
        pa.dis[i] = (i < d->ndims) ? d->dim_is[i] : 0;
        pa.dos[i] = (i < d->ndims) ? d->dim_os[i] : 0;
        pa.dtw[i] = (i < d->ndims) ? d->dim_tw[i] : 0;
    }
    if (bd >= 0) {
        sbase += chunk_adv(d->src_buf, cs, d->dim_is[bd]);
        dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
        pa.dn[bd] = cn;
    }
    pa.src = bufs[d->src_buf] + sbase;
    pa.dst = bufs[d->dst_buf] + dbase;
    pa.src_im = d->src_im;
    pa.dst_im = d->dst_im;
    pa.is_l = d->is_l;
    pa.os_l = d->os_l;
    pa.wL = (d->table >= 0) ? (const cplx *)tables[d->table] : NULL;
    pa.tw_n = d->tw_n;
    pa.tw_shift = d->tw_shift;
    pa.tw_lo = d->tw_n ? (const cplx *)tables[d->tw_lo] : NULL;
    pa.tw_hi = d->tw_n ? (const cplx *)tables[d->tw_hi] : NULL;
    pa.L = d->L;
    pa.nrad = d->nradices;
    for (int i = 0; i < FFTW_AMD_MAX_RADICES; ++i) pa.rad[i] = (i < d->nradices) ? d->radices[i] : 1;
    pa.ndims = d->ndims;
    pa.T = d->tile;
    pa.lo_n = d->tile_lo_n > 1 ? d->tile_lo_n : 1;
    pa.lo_is = d->tile_lo_is; pa.lo_os = d->tile_lo_os;
    while (pa.T > pa.lo_n && (i64)d->L * (pa.T | 1) > 5120) pa.T -= pa.lo_n;   /* tile of a tuned variant may not fit here */
    if (pa.T < pa.lo_n) pa.T = pa.lo_n;
    pa.T -= pa.T % pa.lo_n;
    pa.ld = (pa.T > 1) ? (pa.T | 1) : 1;
    pa.flags = d->flags;
    pa.ntiles = (pa.dn[0] + pa.T / pa.lo_n - 1) / (pa.T / pa.lo_n);
    /* a stride-0 dim cannot be the coalescing index */
    pa.in_t_fast = (pa.dn[0] > 1 && iabs64(pa.dis[0]) <= iabs64(pa.is_l)) || pa.L == 1;
    pa.out_t_fast = (pa.dn[0] > 1 && iabs64(pa.dos[0]) <= iabs64(pa.os_l)) || pa.L == 1;

    i64 nblocks = pa.ntiles;
    for (int i = 1; i < d->ndims; ++i) nblocks *= pa.dn[i];
    if (nblocks <= 0) return 0;
    dim3 grid;
    if (nblocks <= 0x7fffffffLL) grid = dim3((unsigned)nblocks, 1, 1);
    else {
        /* split over y; the kernel recombines.  nblocks must factor: use 65535-ish rows */
        unsigned gy = (unsigned)((nblocks + 0x3fffffffLL) / 0x40000000LL);
        while (nblocks % gy) ++gy;
        grid = dim3((unsigned)(nblocks / gy), gy, 1);
    }
    size_t lds = (size_t)2 * pa.L * pa.ld * sizeof(cplx);
    if (lds > 160 * 1024) {
        fprintf(stderr, "fftw3_amd: internal error: pass tile needs %zu B of LDS\n", lds);
        return -1;
    }
    if (fa_attr_needed(g_lds_attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass_generic_kernel<false, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FA_CHECK(hipFuncSetAttribute((const void *)pass_generic_kernel<true, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FA_CHECK(hipFuncSetAttribute((const void *)pass_generic_kernel<false, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FA_CHECK(hipFuncSetAttribute((const void *)pass_generic_kernel<true, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        fa_attr_set(g_lds_attr_done);
    }
    /* 16-byte vector access when the element is an aligned interleaved pair */
    bool vin = d->src_im == 1 && !(d->flags & FFTW_AMD_F_REAL_IN) &&
               ((uintptr_t)pa.src % 16 == 0) && (pa.is_l % 2 == 0);
    bool vout = d->dst_im == 1 && !(d->flags & FFTW_AMD_F_REAL_OUT) &&
                ((uintptr_t)pa.dst % 16 == 0) && (pa.os_l % 2 == 0);
    for (int i = 0; i < d->ndims; ++i) {
        if (pa.dis[i] % 2) vin = false;
        if (pa.dos[i] % 2) vout = false;
    }
    if (pa.lo_is % 2) vin = false;
    if (pa.lo_os % 2) vout = false;
    {
        /* all radices have register butterflies and the tile is at most 4096
           elements: the single-image kernel (two workgroups per CU) */
        static int inplace_mode = -1;
        bool small_radices = true;
        if (inplace_mode < 0) { const char *e = getenv("FFTW_AMD_INPLACE"); inplace_mode = e ? atoi(e) : 1; }
        for (int i = 0; i < d->nradices; ++i)
            if (d->radices[i] > 16 || d->radices[i] == 6 || d->radices[i] == 9 || d->radices[i] == 10 ||
                d->radices[i] == 12 || d->radices[i] == 14 || d->radices[i] == 15) small_radices = false;
        /* ping-pong images that fit twice on a CU need no help; larger tiles take the single image */
        if (inplace_mode && small_radices && (i64)pa.L * pa.T <= 4096 && lds > 80 * 1024) {
            static std::atomic<unsigned> done{0};
            size_t lds1 = (size_t)pa.L * pa.ld * sizeof(cplx);
            if (fa_attr_needed(done)) {
                FA_CHECK(hipFuncSetAttribute((const void *)pass_inplace_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                FA_CHECK(hipFuncSetAttribute((const void *)pass_inplace_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                FA_CHECK(hipFuncSetAttribute((const void *)pass_inplace_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                FA_CHECK(hipFuncSetAttribute((const void *)pass_inplace_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                fa_attr_set(done);
            }
            if (vin && vout) hipLaunchKernelGGL((pass_inplace_kernel<true, true>), grid, dim3(256), lds1, st, pa);
            else if (vin) hipLaunchKernelGGL((pass_inplace_kernel<true, false>), grid, dim3(256), lds1, st, pa);
            else if (vout) hipLaunchKernelGGL((pass_inplace_kernel<false, true>), grid, dim3(256), lds1, st, pa);
            else hipLaunchKernelGGL((pass_inplace_kernel<false, false>), grid, dim3(256), lds1, st, pa);
            return 0;
        }
    }
    if (vin && vout) launch_pass_variant<true, true>(pa, grid, lds, st);
    else if (vin) launch_pass_variant<true, false>(pa, grid, lds, st);
    else if (vout) launch_pass_variant<false, true>(pa, grid, lds, st);
    else launch_pass_variant<false, false>(pa, grid, lds, st);
    return 0;
}

/* index space of an element-wise step (ElemIdx): dims with the chunk applied, the
   transform index of extent K at position kpos.  Returns 0 when there is nothing to do.
   If the inner part would not fit 32 bits the transform index simply goes first. */
static int elem_fill(ElemIdx *e, const fftw_amd_step_desc *d, i64 K, i64 cn, int kpos, dim3 *grid) {
    int bd = d->batch_dim;
    for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) {
        e->dn[i] = (i < d->ndims) ? d->dim_n[i] : 1;
        e->dis[i] = (i < d->ndims) ? d->dim_is[i] : 0;
        e->dos[i] = (i < d->ndims) ? d->dim_os[i] : 0;
    }
    if (bd >= 0) e->dn[bd] = cn;
    e->ndims = d->ndims;
    if (kpos < 0) kpos = 0;
    if (kpos > d->ndims) kpos = d->ndims;
    if (bd >= 0 && kpos > bd) kpos = bd;          /* the batch loop stays outside */
    for (;;) {
        unsigned long long inner = (unsigned long long)K;
        bool ok = K > 0 && K < 0x7fffffffLL;
        for (int i = 0; i < kpos && ok; ++i) {
            if (e->dn[i] <= 0 || e->dn[i] >= 0x7fffffffLL) { ok = false; break; }
            inner *= (unsigned long long)e->dn[i];
            if (inner >= 0xffffff00ULL) ok = false;
        }
        if (ok) { e->inner = (unsigned)inner; break; }
        if (kpos == 0) return 0;                  /* K itself out of range: nothing sane to launch */
        kpos = 0;
    }
    e->kpos = kpos;
    e->K = (unsigned)K;
    e->nblk = (e->inner + 255u) / 256u;
    i64 nvb = e->nblk;
    for (int i = kpos; i < d->ndims; ++i) {
        if (e->dn[i] <= 0) return 0;
        nvb *= e->dn[i];
    }
    e->nvb = nvb;
    i64 blocks = nvb;
    if (blocks > 256 * 64) blocks = 256 * 64;     /* the kernels loop over virtual blocks beyond that */
    *grid = dim3((unsigned)blocks, 1, 1);
    return nvb > 0;
}

static int fill_copy_args(CopyArgs *ca, const fftw_amd_step_desc *d, double *const *bufs,
                          void *const *tables, i64 cs, i64 cn, dim3 *grid) {
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    if (bd >= 0) {
        sbase += chunk_adv(d->src_buf, cs, d->dim_is[bd]);
        dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
    }
    ca->src = bufs[d->src_buf] + sbase;
    ca->dst = bufs[d->dst_buf] + dbase;
    ca->src_im = d->src_im;
    ca->dst_im = d->dst_im;
    ca->is_k = d->is_l;
    ca->os_k = d->os_l;
    ca->K = d->aux_n;
    ca->Kvalid = d->aux_valid;
    ca->flags = d->flags;
    ca->tab = (d->table >= 0) ? (const cplx *)tables[d->table] : NULL;
    ca->perm = (d->table2 >= 0) ? (const i64 *)tables[d->table2] : NULL;
    return elem_fill(&ca->e, d, d->aux_n, cn, 0, grid);
}

/* r2r length n from the inner real-DFT length N of a fused step */
static i64 r2r_len_of(int mode, i64 N) {
    if (mode == FFTW_AMD_R2R_POST_E00) return N / 2 + 1;
    if (mode == FFTW_AMD_R2R_POST_O00) return N / 2 - 1;
    return N;
}

static int launch_step_kind(const fftw_amd_step_desc *d, double *const *bufs,
                            void *const *tables, long long cs, long long cn, hipStream_t st);

/* A launch that the runtime refuses (bad grid / LDS configuration, missing code object) leaves
   no trace unless hipGetLastError is asked: without this check the step would "succeed" and the
   output would silently keep its old contents. */
extern "C" int fa_hip_launch_step(const fftw_amd_step_desc *d, double *const *bufs,
                                  void *const *tables, long long cs, long long cn,
                                  void *stream) {
    int r = launch_step_kind(d, bufs, tables, cs, cn, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "fftw3_amd: kernel launch of step kind %d (L=%d, variant %d) failed: %s\n",
                d->kind, d->L, d->variant, hipGetErrorString(e));
        return -1;
    }
    return r;
}

static int launch_step_kind(const fftw_amd_step_desc *d, double *const *bufs,
                            void *const *tables, long long cs, long long cn, hipStream_t st) {
    switch (d->kind) {
    case FFTW_AMD_STEP_PASS:
        return launch_pass(d, bufs, tables, cs, cn, st);
    case FFTW_AMD_STEP_COPY:
    case FFTW_AMD_STEP_HERM_EXPAND: {
        CopyArgs ca;
        dim3 grid;
        if (!fill_copy_args(&ca, d, bufs, tables, cs, cn, &grid)) return 0;
        if (d->kind == FFTW_AMD_STEP_COPY) {
            if (d->flags & (FFTW_AMD_F_PERM_SRC | FFTW_AMD_F_PERM_DST)) {
                if (grid.x >= 8) grid.x &= ~7u;          /* rows stay on one XCD: a grid of whole rounds of the eight */
                hipLaunchKernelGGL(copy_kernel<1>, grid, dim3(256), 0, st, ca);
            } else {
                const i64 ngroups = (ca.e.nvb + 3) / 4;
                if ((i64)grid.x > ngroups) grid.x = (unsigned)ngroups;
                hipLaunchKernelGGL(copy_kernel<4>, grid, dim3(256), 0, st, ca);
            }
        }
        else
            hipLaunchKernelGGL(herm_expand_kernel, grid, dim3(256), 0, st, ca);
        return 0;
    }
    case FFTW_AMD_STEP_R2C_POST:
    case FFTW_AMD_STEP_C2R_PRE: {
        RealArgs ra;
        int bd = d->batch_dim;
        i64 sbase = d->src_base, dbase = d->dst_base;
        if (bd >= 0) {
            sbase += chunk_adv(d->src_buf, cs, d->dim_is[bd]);
            dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
        }
        ra.src = bufs[d->src_buf] + sbase;
        ra.dst = bufs[d->dst_buf] + dbase;
        ra.src_im = d->src_im;
        ra.dst_im = d->dst_im;
        ra.is_k = d->is_l;
        ra.os_k = d->os_l;
        ra.h = d->aux_n / 2;
        ra.npair = ra.h / 2 + 1;
        ra.r2r = d->variant;
        ra.twmul = d->tile > 0 ? d->tile : 1;
        ra.rn = r2r_len_of(d->variant, d->aux_n);
        ra.tw_lo = (const cplx *)tables[d->tw_lo];
        ra.tw_hi = (const cplx *)tables[d->tw_hi];
        ra.tw_shift = d->tw_shift;
        ra.flags = d->flags;
        if (d->kind == FFTW_AMD_STEP_R2C_POST && (ra.r2r == FFTW_AMD_R2R_POST_E10 || ra.r2r == FFTW_AMD_R2R_POST_O10) &&
            ra.twmul == 4 && ra.src_im == 1 && ra.is_k == 2 && ra.os_k == 1 && ra.rn == 2 * ra.h && ra.h >= 4 && ra.h % 2 == 0 &&
            ra.rn < (1LL << 31) && d->ndims == 1 && bd == 0 && d->kpos == 0 && cn > 0 && cn < 65536 &&
            (d->dim_is[0] % 2) == 0 && ((uintptr_t)ra.src % 16) == 0 &&
            !(d->flags & (FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_SWAP_OUT | FFTW_AMD_F_CONJ_OUT))) {
            /* DCT-II / DST-II: streaming untangle + epilogue (dct2_untangle_fast_kernel) */
            DctFast f;
            f.src = ra.src; f.dst = ra.dst; f.sbatch = d->dim_is[0]; f.dbatch = d->dim_os[0];
            f.n = (unsigned)ra.rn; f.nitems = (unsigned)ra.npair;
            f.tw_lo = ra.tw_lo; f.tw_hi = ra.tw_hi; f.tw_shift = ra.tw_shift;
            f.odd = ra.r2r == FFTW_AMD_R2R_POST_O10;
            dim3 g((f.nitems + 255) / 256, (unsigned)cn, 1);
            hipLaunchKernelGGL(dct2_untangle_fast_kernel<false>, g, dim3(256), 0, st, f);
            return 0;
        }
        if (d->kind == FFTW_AMD_STEP_C2R_PRE && (ra.r2r == FFTW_AMD_R2R_PRE_E01 || ra.r2r == FFTW_AMD_R2R_PRE_O01) &&
            ra.twmul == 4 && ra.dst_im == 1 && ra.os_k == 2 && ra.is_k == 1 && ra.rn == 2 * ra.h && ra.h >= 4 && ra.h % 2 == 0 &&
            ra.rn < (1LL << 31) && d->ndims == 1 && bd == 0 && d->kpos == 0 && cn > 0 && cn < 65536 &&
            (d->dim_os[0] % 2) == 0 && ((uintptr_t)ra.dst % 16) == 0 &&
            !(d->flags & (FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_SWAP_OUT | FFTW_AMD_F_CONJ_OUT | FFTW_AMD_F_REAL_OUT))) {
            /* DCT-III / DST-III: streaming prologue + tangle (dct3_tangle_fast_kernel) */
            DctFast f;
            f.src = ra.src; f.dst = ra.dst; f.sbatch = d->dim_is[0]; f.dbatch = d->dim_os[0];
            f.n = (unsigned)ra.rn; f.nitems = (unsigned)ra.npair;
            f.tw_lo = ra.tw_lo; f.tw_hi = ra.tw_hi; f.tw_shift = ra.tw_shift;
            f.odd = ra.r2r == FFTW_AMD_R2R_PRE_O01;
            dim3 g((f.nitems + 255) / 256, (unsigned)cn, 1);
            hipLaunchKernelGGL(dct3_tangle_fast_kernel<false>, g, dim3(256), 0, st, f);
            return 0;
        }
        {
            /* streaming form for the layout of the large 1-D plans (see r2c_post_fast_kernel) */
            const bool r2c = d->kind == FFTW_AMD_STEP_R2C_POST;
            const int swap_mask = FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_SWAP_OUT | FFTW_AMD_F_CONJ_OUT | FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT;
            if (ra.r2r == 0 && ra.twmul == 1 && ra.src_im == 1 && ra.dst_im == 1 && ra.is_k == 2 && ra.os_k == 2 &&
                !(d->flags & swap_mask) && d->ndims == 1 && bd == 0 && d->kpos == 0 && ra.h >= 2 && ra.h < (1LL << 30) &&
                cn > 0 && cn < 65536 && (d->dim_is[0] % 2) == 0 && (d->dim_os[0] % 2) == 0 &&
                ((uintptr_t)ra.src % 16) == 0 && ((uintptr_t)ra.dst % 16) == 0) {
                Real2Fast f2;
                f2.src = ra.src; f2.dst = ra.dst;
                f2.sbatch = d->dim_is[0]; f2.dbatch = d->dim_os[0];
                f2.h = (unsigned)ra.h; f2.npair = (unsigned)ra.npair;
                f2.tw_lo = ra.tw_lo; f2.tw_hi = ra.tw_hi; f2.tw_shift = ra.tw_shift;
                dim3 g((f2.npair + 255) / 256, (unsigned)cn, 1);
                if (r2c) {
                    if (d->flags & FFTW_AMD_F_NT_OUT) hipLaunchKernelGGL(r2c_post_fast_kernel<true>, g, dim3(256), 0, st, f2);
                    else hipLaunchKernelGGL(r2c_post_fast_kernel<false>, g, dim3(256), 0, st, f2);
                } else {
                    if (d->flags & FFTW_AMD_F_NT_IN) hipLaunchKernelGGL(c2r_pre_fast_kernel<true>, g, dim3(256), 0, st, f2);
                    else hipLaunchKernelGGL(c2r_pre_fast_kernel<false>, g, dim3(256), 0, st, f2);
                }
                return 0;
            }
        }
        dim3 grid;
        if (!elem_fill(&ra.e, d, ra.npair, cn, d->kpos, &grid)) return 0;
        if (d->kind == FFTW_AMD_STEP_R2C_POST)
            hipLaunchKernelGGL(r2c_post_kernel, grid, dim3(256), 0, st, ra);
        else
            hipLaunchKernelGGL(c2r_pre_kernel, grid, dim3(256), 0, st, ra);
        return 0;
    }
    case FFTW_AMD_STEP_R2C_POST4:
    case FFTW_AMD_STEP_C2R_PRE4: {
        Real4Args ra;
        int bd = d->batch_dim;
        i64 sbase = d->src_base, dbase = d->dst_base;
        if (bd >= 0) {
            sbase += chunk_adv(d->src_buf, cs, d->dim_is[bd]);
            dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
        }
        ra.src = bufs[d->src_buf] + sbase;
        ra.dst = bufs[d->dst_buf] + dbase;
        ra.src_im = d->src_im;
        ra.dst_im = d->dst_im;
        ra.is_k = d->is_l;
        ra.os_k = d->os_l;
        ra.vs = d->aux_valid;            /* distance between the two quarter-length vectors */
        ra.m = d->aux_n / 4;
        ra.r2r = d->variant;
        ra.twmul = d->tile > 0 ? d->tile : 1;
        ra.rn = r2r_len_of(d->variant, d->aux_n);
        ra.npair = (d->kind == FFTW_AMD_STEP_R2C_POST4) ? ra.m / 2 + 1 : ra.m / 2 + 1;
        ra.tw_lo = (const cplx *)tables[d->tw_lo];
        ra.tw_hi = (const cplx *)tables[d->tw_hi];
        ra.tw_shift = d->tw_shift;
        ra.flags = d->flags;
        {
            /* streaming form for the layout of the large 1-D plans (see r2c_post4_fast_kernel) */
            const bool r2c = d->kind == FFTW_AMD_STEP_R2C_POST4;
            const i64 zs = r2c ? ra.is_k : ra.os_k, ys = r2c ? ra.os_k : ra.is_k;   /* Z side / Y side strides */
            const int swap_mask = FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_SWAP_OUT | FFTW_AMD_F_CONJ_OUT | FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT;
            if (ra.r2r == 0 && ra.twmul == 1 && ra.src_im == 1 && ra.dst_im == 1 && zs == 4 && ys == 2 && ra.vs == 2 &&
                !(d->flags & swap_mask) && d->ndims == 1 && bd == 0 && d->kpos == 0 && ra.m >= 2 && ra.m < (1LL << 30) &&
                cn > 0 && cn < 65536 && (d->dim_is[0] % 2) == 0 && (d->dim_os[0] % 2) == 0 &&
                ((uintptr_t)ra.src % 16) == 0 && ((uintptr_t)ra.dst % 16) == 0) {
                Real4Fast fa4;
                fa4.src = ra.src; fa4.dst = ra.dst;
                fa4.sbatch = d->dim_is[0]; fa4.dbatch = d->dim_os[0];
                fa4.m = (unsigned)ra.m; fa4.npair = (unsigned)ra.npair;
                fa4.tw_lo = ra.tw_lo; fa4.tw_hi = ra.tw_hi; fa4.tw_shift = ra.tw_shift;
                fa4.nt_out = 0;
                dim3 g((fa4.npair + 255) / 256, (unsigned)cn, 1);
                if (r2c) {
                    if (d->flags & FFTW_AMD_F_NT_OUT) hipLaunchKernelGGL(r2c_post4_fast_kernel<true>, g, dim3(256), 0, st, fa4);
                    else hipLaunchKernelGGL(r2c_post4_fast_kernel<false>, g, dim3(256), 0, st, fa4);
                } else {
                    if (d->flags & FFTW_AMD_F_NT_IN) hipLaunchKernelGGL(c2r_pre4_fast_kernel<true>, g, dim3(256), 0, st, fa4);
                    else hipLaunchKernelGGL(c2r_pre4_fast_kernel<false>, g, dim3(256), 0, st, fa4);
                }
                return 0;
            }
        }
        dim3 grid;
        if (!elem_fill(&ra.e, d, ra.npair, cn, d->kpos, &grid)) return 0;
        if (d->kind == FFTW_AMD_STEP_R2C_POST4)
            hipLaunchKernelGGL(r2c_post4_kernel, grid, dim3(256), 0, st, ra);
        else
            hipLaunchKernelGGL(c2r_pre4_kernel, grid, dim3(256), 0, st, ra);
        return 0;
    }
    case FFTW_AMD_STEP_R2R: {
        R2RArgs ra;
        int bd = d->batch_dim;
        i64 sbase = d->src_base, dbase = d->dst_base;
        if (bd >= 0) {
            sbase += chunk_adv(d->src_buf, cs, d->dim_is[bd]);
            dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
        }
        ra.src = bufs[d->src_buf] + sbase;
        ra.dst = bufs[d->dst_buf] + dbase;
        ra.src_im = d->src_im;
        ra.dst_im = d->dst_im;
        ra.is_k = d->is_l;
        ra.os_k = d->os_l;
        ra.n = d->aux_n;
        ra.K = d->aux_valid;
        ra.tw_lo = (d->tw_lo >= 0) ? (const cplx *)tables[d->tw_lo] : NULL;
        ra.tw_hi = (d->tw_hi >= 0) ? (const cplx *)tables[d->tw_hi] : NULL;
        ra.tw_shift = d->tw_shift;
        ra.mode = d->variant;
        if ((ra.mode == FFTW_AMD_R2R_PRE_E10 || ra.mode == FFTW_AMD_R2R_PRE_O10) && ra.is_k == 1 && ra.os_k == 2 &&
            ra.dst_im == 1 && ra.n >= 8 && ra.n % 4 == 0 && ra.n < (1LL << 31) && d->ndims == 1 && bd == 0 && d->kpos == 0 &&
            cn > 0 && cn < 65536 && (d->dim_is[0] % 2) == 0 && (d->dim_os[0] % 2) == 0 &&
            ((uintptr_t)ra.src % 16) == 0 && ((uintptr_t)ra.dst % 16) == 0) {
            DctFast f;
            f.src = ra.src; f.dst = ra.dst; f.sbatch = d->dim_is[0]; f.dbatch = d->dim_os[0];
            f.n = (unsigned)ra.n; f.nitems = (unsigned)(ra.n / 4);
            f.tw_lo = NULL; f.tw_hi = NULL; f.tw_shift = 0;
            f.odd = ra.mode == FFTW_AMD_R2R_PRE_O10;
            dim3 g((f.nitems + 255) / 256, (unsigned)cn, 1);
            if (d->flags & FFTW_AMD_F_NT_IN) hipLaunchKernelGGL(dct2_shuffle_fast_kernel<true>, g, dim3(256), 0, st, f);
            else hipLaunchKernelGGL(dct2_shuffle_fast_kernel<false>, g, dim3(256), 0, st, f);
            return 0;
        }
        if ((ra.mode == FFTW_AMD_R2R_POST_E01 || ra.mode == FFTW_AMD_R2R_POST_O01) && ra.is_k == 2 && ra.src_im == 1 &&
            ra.os_k == 1 && ra.n >= 8 && ra.n % 4 == 0 && ra.n < (1LL << 31) && d->ndims == 1 && bd == 0 && d->kpos == 0 &&
            cn > 0 && cn < 65536 && (d->dim_is[0] % 2) == 0 && (d->dim_os[0] % 2) == 0 &&
            ((uintptr_t)ra.src % 16) == 0 && ((uintptr_t)ra.dst % 16) == 0) {
            DctFast f;
            f.src = ra.src; f.dst = ra.dst; f.sbatch = d->dim_is[0]; f.dbatch = d->dim_os[0];
            f.n = (unsigned)ra.n; f.nitems = (unsigned)(ra.n / 4);
            f.tw_lo = NULL; f.tw_hi = NULL; f.tw_shift = 0;
            f.odd = ra.mode == FFTW_AMD_R2R_POST_O01;
            dim3 g((f.nitems + 255) / 256, (unsigned)cn, 1);
            if (d->flags & FFTW_AMD_F_NT_OUT) hipLaunchKernelGGL(dct3_unshuffle_fast_kernel<true>, g, dim3(256), 0, st, f);
            else hipLaunchKernelGGL(dct3_unshuffle_fast_kernel<false>, g, dim3(256), 0, st, f);
            return 0;
        }
        dim3 grid;
        if (!elem_fill(&ra.e, d, ra.K, cn, d->kpos, &grid)) return 0;
        hipLaunchKernelGGL(r2r_kernel, grid, dim3(256), 0, st, ra);
        return 0;
    }
    case FFTW_AMD_STEP_RADER_MUL: {
        RaderArgs ra;
        int bd = d->batch_dim;
        i64 dbase = d->dst_base;
        i64 nvec = 1;
        for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) {
            ra.dn[i] = (i < d->ndims) ? d->dim_n[i] : 1;
            ra.dos[i] = (i < d->ndims) ? d->dim_os[i] : 0;
        }
        if (bd >= 0) {
            dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
            ra.dn[bd] = cn;
        }
        for (int i = 0; i < d->ndims; ++i) nvec *= ra.dn[i];
        ra.work = bufs[d->src_buf] + d->src_base;
        ra.x0 = bufs[d->aux_buf] + d->aux_base;
        ra.dst = bufs[d->dst_buf] + dbase;
        ra.dst_im = d->dst_im;
        ra.pm1 = d->aux_n;
        ra.nvec = nvec;
        ra.total = nvec * ra.pm1;
        ra.omega = (const cplx *)tables[d->table];
        ra.ndims = d->ndims;
        ra.flags = d->flags;
        if (ra.total <= 0) return 0;
        dim3 grid;
        grid_for(ra.total, &grid);
        hipLaunchKernelGGL(rader_mul_kernel, grid, dim3(256), 0, st, ra);
        return 0;
    }
    default:
        fprintf(stderr, "fftw3_amd: unknown step kind %d\n", d->kind);
        return -1;
    }
}
