/*
 * kernels_r1.hip -- instantiations and launcher of the one-stage rows kernel (pass1r.hpp):
 * dense rows of R = 2 ... 32 points, one butterfly per row.  A translation unit of its own.
 */
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "pass3s.hpp"
#include "pass1r.hpp"

#define FA_R1_LENGTHS(X) \
    X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) \
    X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)

template <int R>
static void launch_1r(const P3SArgs &pa, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    const size_t lds = P1RGeom<R>::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass1r_kernel<R>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL((pass1r_kernel<R>), grid, dim3(256), lds, st, pa);
}

/* rows per tile of the one-stage rows kernel for length L (0: none) */
extern "C" int fa_hip_r1_tile(int L) {
    switch (L) {
#define X(R_) case R_: return P1RGeom<R_>::T;
        FA_R1_LENGTHS(X)
#undef X
    }
    return 0;
}

/* dense interleaved rows of a short length in one butterfly per row; 1 = not applicable (the
   caller falls back to the two-stage or the LDS kernel) */
int fa_launch_pass1r(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                     i64 cs, i64 cn, hipStream_t st) {
    P3SArgs pa;
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    const int T = fa_hip_r1_tile(d->L);
    (void)tables;
    if (T <= 0 || d->tile != T || d->src_im != 1 || d->dst_im != 1 || d->tw_n ||
        d->is_l != 2 || d->os_l != 2 || d->tile_lo_n > 1 || d->ndims < 1 ||
        d->dim_is[0] != 2 * (i64)d->L || d->dim_os[0] != 2 * (i64)d->L ||
        (d->flags & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT | FFTW_AMD_F_CONJ_OUT)))
        return 1;
    for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) {
        pa.dn[i] = (i < d->ndims) ? d->dim_n[i] : 1;
        pa.dis[i] = (i < d->ndims) ? d->dim_is[i] : 0;
        pa.dos[i] = (i < d->ndims) ? d->dim_os[i] : 0;
    }
    if (bd >= 0) {
        sbase += chunk_adv(d->src_buf, cs, d->dim_is[bd]);
        dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
        pa.dn[bd] = cn;
    }
    pa.src = bufs[d->src_buf] + sbase;
    pa.dst = bufs[d->dst_buf] + dbase;
    if (((uintptr_t)pa.src % 16) || ((uintptr_t)pa.dst % 16)) return 1;
    for (int i = 0; i < d->ndims; ++i)
        if ((pa.dis[i] % 2) || (pa.dos[i] % 2)) return 1;
    pa.wL = NULL;
    pa.ndims = d->ndims;
    pa.flags = d->flags;
    pa.ntiles = (pa.dn[0] + T - 1) / T;
    i64 nblocks = pa.ntiles;
    for (int i = 1; i < d->ndims; ++i) nblocks *= pa.dn[i];
    if (nblocks <= 0) return 0;
    if (nblocks > 0x7fffffffLL) return 1;
    dim3 grid((unsigned)nblocks, 1, 1);
    switch (d->L) {
#define X(R_) case R_: launch_1r<R_>(pa, grid, st); return 0;
        FA_R1_LENGTHS(X)
#undef X
    }
    return 1;
}

/* ---- dense real rows of n = 2L = 4 ... 64 points <-> half spectra (pass1r_real_kernel) ------------------------ */

template <int R>
static void launch_1r_real(const P3SArgs &pa, dim3 grid, hipStream_t st, bool inverse, bool pad) {
    static std::atomic<unsigned> attr_done{0};
    const size_t lds = P1RRealGeom<R>::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass1r_real_kernel<R, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FA_CHECK(hipFuncSetAttribute((const void *)pass1r_real_kernel<R, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FA_CHECK(hipFuncSetAttribute((const void *)pass1r_real_kernel<R, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FA_CHECK(hipFuncSetAttribute((const void *)pass1r_real_kernel<R, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    if (inverse && pad) hipLaunchKernelGGL((pass1r_real_kernel<R, false, true>), grid, dim3(256), lds, st, pa);
    else if (inverse) hipLaunchKernelGGL((pass1r_real_kernel<R, false, false>), grid, dim3(256), lds, st, pa);
    else if (pad) hipLaunchKernelGGL((pass1r_real_kernel<R, true, true>), grid, dim3(256), lds, st, pa);
    else hipLaunchKernelGGL((pass1r_real_kernel<R, true, false>), grid, dim3(256), lds, st, pa);
}

extern "C" int fa_hip_r2c_rows1_tile(int L) { return fa_hip_r1_tile(L); }

/* The step (FFTW_AMD_F_R2C_ROWS / _C2R_ROWS with L <= 32) has no other executor; the planner emits it only when a
   loop of at least 256 rows has exactly the dense strides, so anything else is a caller error and fails loudly. */
int fa_launch_r2crows1(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                       i64 cs, i64 cn, hipStream_t st) {
    P3SArgs pa;
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    const int T = fa_hip_r2c_rows1_tile(d->L);
    const bool inverse = (d->flags & FFTW_AMD_F_C2R_ROWS) != 0;
    const i64 rs = 2 * (i64)d->L, cst = 2 * ((i64)d->L + 1);          /* row pitch on the real / complex side */
    const i64 want_is = inverse ? cst : rs, want_os = inverse ? rs : cst;
    int dense = -1;
    bool pad = false;
    if (T <= 0 || d->tile != T || d->src_im != 1 || d->dst_im != 1 || d->is_l != 2 || d->os_l != 2 ||
        d->aux_valid || d->aux_buf > 0 || d->tile_lo_n > 1) {
        fprintf(stderr, "fftw3_amd: internal error: short real rows step with an unsupported layout\n");
        abort();
    }
    for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) {
        pa.dn[i] = (i < d->ndims) ? d->dim_n[i] : 1;
        pa.dis[i] = (i < d->ndims) ? d->dim_is[i] : 0;
        pa.dos[i] = (i < d->ndims) ? d->dim_os[i] : 0;
    }
    if (bd >= 0) {
        sbase += chunk_adv(d->src_buf, cs, d->dim_is[bd]);
        dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
        pa.dn[bd] = cn;
    }
    /* the dense loop becomes the tile dim (the planner's tile dim is whichever loop has the smallest stride) */
    for (int i = 0; i < d->ndims; ++i)
        if (pa.dis[i] == want_is && pa.dos[i] == want_os && (dense < 0 || pa.dn[i] > pa.dn[dense])) dense = i;
    if (dense < 0) {
        /* FFTW's padded layout: both sides advance by 2 (L + 1) doubles per row */
        for (int i = 0; i < d->ndims; ++i)
            if (pa.dis[i] == cst && pa.dos[i] == cst && (dense < 0 || pa.dn[i] > pa.dn[dense])) dense = i;
        pad = dense >= 0;
    }
    pa.src = bufs[d->src_buf] + sbase;
    pa.dst = bufs[d->dst_buf] + dbase;
    bool bad = dense < 0 || ((uintptr_t)pa.src % 16) || ((uintptr_t)pa.dst % 16);
    for (int i = 0; i < d->ndims; ++i)
        if ((pa.dis[i] % 2) || (pa.dos[i] % 2)) bad = true;
    if (bad) {
        fprintf(stderr, "fftw3_amd: fftw_execute_dft_r2c / _c2r needs arrays laid out and aligned like the ones the plan "
                        "was created with\n");
        abort();
    }
    if (dense != 0) {
        i64 t;
        t = pa.dn[0]; pa.dn[0] = pa.dn[dense]; pa.dn[dense] = t;
        t = pa.dis[0]; pa.dis[0] = pa.dis[dense]; pa.dis[dense] = t;
        t = pa.dos[0]; pa.dos[0] = pa.dos[dense]; pa.dos[dense] = t;
    }
    pa.wL = NULL;
    pa.tw_lo = (const cplx *)tables[d->tw_lo];
    pa.tw_hi = (const cplx *)tables[d->tw_hi];
    pa.tw_shift = d->tw_shift;
    pa.ndims = d->ndims;
    pa.flags = 0;
    pa.ntiles = (pa.dn[0] + T - 1) / T;
    i64 nblocks = pa.ntiles;
    for (int i = 1; i < d->ndims; ++i) nblocks *= pa.dn[i];
    if (nblocks <= 0) return 0;
    if (nblocks > 0x7fffffffLL) {
        fprintf(stderr, "fftw3_amd: short real rows step with more than 2^31 tiles\n");
        abort();
    }
    dim3 grid((unsigned)nblocks, 1, 1);
    switch (d->L) {
#define X(R_) case R_: launch_1r_real<R_>(pa, grid, st, inverse, pad); return 0;
        FA_R1_LENGTHS(X)
#undef X
    }
    fprintf(stderr, "fftw3_amd: internal error: no short real rows kernel for half length %d\n", d->L);
    abort();
}
