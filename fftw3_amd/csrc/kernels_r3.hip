/*
 * kernels_r3.hip -- instantiations and launcher of the general three-stage rows
 * kernels (pass3g.hpp) for every length of r3_menu.inc (X(L, R1, R2, R3) entries chosen
 * by tools/gen_r3_menu.py).  A translation unit of its own: it compiles in parallel with
 * kernels.hip and kernels_rr.hip.
 */
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "pass3s.hpp"
#include "pass3g.hpp"
#include "r2crows.hpp"

template <int R1, int R2, int R3>
static void launch_3g(const P3SArgs &pa, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    static_assert(P3GGeom<R1, R2, R3>::fits, "menu entry exceeds the per-item element budget");
    const size_t lds = P3GGeom<R1, R2, R3>::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass3g_kernel<R1, R2, R3>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL((pass3g_kernel<R1, R2, R3>), grid, dim3(256), lds, st, pa);
}

extern "C" int fa_hip_r3w_has(int L);              /* kernels_r3w.hip */

/* rows per tile of the three-stage kernel for length L (0: none).  2048 / 4096 / 8192 are the
   tuned pass3s kernels of kernels_rr.hip. */
extern "C" int fa_hip_r3_tile(int L) {
    if (L == 2048 || L == 4096 || L == 8192) return 8192 / L;
    if (L == 16384) return 1;                      /* pass3w.hpp: one 512-item workgroup per row */
    if (L > 8192 && fa_hip_r3w_has(L)) return 1;   /* kernels_r3w.hip: wide three-stage kernels, one row per 512-item workgroup */
    switch (L) {
#define X(L_, R1_, R2_, R3_) case L_: return P3GGeom<R1_, R2_, R3_>::T;
#include "r3_menu.inc"
#undef X
    }
    return 0;
}

/* contiguous rows of a menu length in one pass; 1 = not applicable */
int fa_launch_pass3g(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                     i64 cs, i64 cn, hipStream_t st) {
    P3SArgs pa;
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    const int T = fa_hip_r3_tile(d->L);
    if (T <= 0 || d->L == 2048 || d->L == 4096 || d->L == 8192 || d->L == 16384 || (d->L > 8192 && fa_hip_r3w_has(d->L)) || d->tile != T || d->src_im != 1 || d->dst_im != 1 ||
        d->tw_n || d->is_l != 2 || d->os_l != 2 || d->tile_lo_n > 1 ||
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
    pa.wL = (const cplx *)tables[d->table];
    pa.ndims = d->ndims;
    pa.flags = d->flags;
    pa.ntiles = (pa.dn[0] + T - 1) / T;
    i64 nblocks = pa.ntiles;
    for (int i = 1; i < d->ndims; ++i) nblocks *= pa.dn[i];
    if (nblocks <= 0) return 0;
    if (nblocks > 0x7fffffffLL) return 1;
    dim3 grid((unsigned)nblocks, 1, 1);
    switch (d->L) {
#define X(L_, R1_, R2_, R3_) case L_: launch_3g<R1_, R2_, R3_>(pa, grid, st); return 0;
#include "r3_menu.inc"
#undef X
    }
    return 1;
}

/* ---- strided / transposed forms (pass3t_kernel), menu r3t_menu.inc ---------------- */

template <int R1, int R2, int R3, bool IN_T, int TW>
static void launch_3t_variant(const P1024Args &pa, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    static_assert(P3TGeom<R1, R2, R3>::fits, "menu entry exceeds the per-item element budget");
    const size_t lds = P3TGeom<R1, R2, R3>::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass3t_kernel<R1, R2, R3, IN_T, TW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL((pass3t_kernel<R1, R2, R3, IN_T, TW>), grid, dim3(256), lds, st, pa);
}

/* column passes (T,T) without / with output twiddle, transposed last passes (L,T) without /
   with input twiddle; anything else goes to the LDS kernel */
template <int R1, int R2, int R3>
static int dispatch_3t(const P1024Args &pa, dim3 grid, hipStream_t st, bool in_t, bool out_t, int tw) {
    if (in_t && out_t) {
        if (tw == 0) { launch_3t_variant<R1, R2, R3, true, 0>(pa, grid, st); return 0; }
        if (tw == 1) { launch_3t_variant<R1, R2, R3, true, 1>(pa, grid, st); return 0; }
        return 1;
    }
    if (!in_t && out_t) {
        if (tw == 0) { launch_3t_variant<R1, R2, R3, false, 0>(pa, grid, st); return 0; }
        if (tw == 2) { launch_3t_variant<R1, R2, R3, false, 2>(pa, grid, st); return 0; }
    }
    return 1;
}

extern "C" int fa_hip_r3tw_tile(int L);            /* kernels_r3tw.hip: the 512-item forms */

/* sequences per tile of the strided three-stage kernel for length L (0: none); lengths with a 512-item form
   (kernels_r3tw.hip) report that form's tile -- it is the one the executor launches */
extern "C" int fa_hip_r3t_tile(int L) {
    if (L > 1024 && fa_hip_r3tw_tile(L) > 0) return fa_hip_r3tw_tile(L);
    switch (L) {
#define X(L_, R1_, R2_, R3_) case L_: return P3TGeom<R1_, R2_, R3_>::T;
#include "r3t_menu.inc"
#undef X
    }
    return 0;
}

int fa_launch_pass3t(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                     i64 cs, i64 cn, hipStream_t st) {
    P1024Args pa;
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    const int T = fa_hip_r3t_tile(d->L);
    if (d->L > 1024 && fa_hip_r3tw_tile(d->L) > 0) return 1;      /* the 512-item form: fa_launch_pass3tw */
    if (T <= 0 || d->tile != T || d->src_im != 1 || d->dst_im != 1 || d->tile_lo_n > 1 ||
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
    pa.lo_sh = 0; pa.lo_is = 0; pa.lo_os = 0;
    pa.ntiles = (pa.dn[0] + T - 1) / T;
    i64 nblocks = pa.ntiles;
    for (int i = 1; i < d->ndims; ++i) nblocks *= pa.dn[i];
    if (nblocks <= 0) return 0;
    if (nblocks > 0x7fffffffLL) return 1;
    if (pa.dn[0] * 4 < T) return 1;              /* a mostly empty tile: the LDS kernel */
    dim3 grid((unsigned)nblocks, 1, 1);
    bool in_t = pa.dn[0] > 1 && iabs64(pa.dis[0]) <= iabs64(pa.is_l);
    bool out_t = pa.dn[0] > 1 && iabs64(pa.dos[0]) <= iabs64(pa.os_l);
    int tw = d->tw_n == 0 ? 0 : ((d->flags & FFTW_AMD_F_TW_IN) ? 2 : 1);
    switch (d->L) {
#define X(L_, R1_, R2_, R3_) case L_: return dispatch_3t<R1_, R2_, R3_>(pa, grid, st, in_t, out_t, tw);
#include "r3t_menu.inc"
#undef X
    }
    return 1;
}

/* ---- fused real rows -> half spectra (r2crows.hpp) ------------------------------- */

template <int R1, int R2>
static void launch_r2cr(const R2CRArgs &ra, dim3 grid, hipStream_t st, bool inverse) {
    static std::atomic<unsigned> attr_done{0};
    const size_t lds = R2CRGeom<R1, R2>::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)r2crows_kernel<R1, R2>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FA_CHECK(hipFuncSetAttribute((const void *)c2rrows_kernel<R1, R2>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    if (inverse) hipLaunchKernelGGL((c2rrows_kernel<R1, R2>), grid, dim3(256), lds, st, ra);
    else hipLaunchKernelGGL((r2crows_kernel<R1, R2>), grid, dim3(256), lds, st, ra);
}

extern "C" int fa_hip_r2c_rows_tile(int L) {
    switch (L) {
    case 64: return R2CRGeom<8, 8>::T;
    case 128: return R2CRGeom<16, 8>::T;
    case 256: return R2CRGeom<16, 16>::T;
    case 512: return R2CRGeom<32, 16>::T;
    case 1024: return R2CRGeom<32, 32>::T;
    }
    return 0;
}

/* A step with FFTW_AMD_F_R2C_ROWS / FFTW_AMD_F_C2R_ROWS has no other executor: the planner only emits it for
   layouts this kernel takes (r2c_rows_layout_ok), so anything else here is a caller error
   (new-array execution with differently aligned arrays) and fails loudly. */
int fa_launch_r2crows1(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                       i64 cs, i64 cn, hipStream_t st);
extern "C" int fa_hip_r2c_rows2m_tile(int L);      /* kernels_r2cm.hip: mixed-radix two-stage lengths, plain r2c / c2r */
int fa_launch_r2crows2m(int L, const R2CRArgs &ra, dim3 grid, hipStream_t st, bool inverse);
extern "C" int fa_hip_r2c_rows3_tile(int L);
int fa_launch_r2crows3(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                       i64 cs, i64 cn, hipStream_t st);

int fa_launch_r2crows(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                      i64 cs, i64 cn, hipStream_t st) {
    /* half lengths above 1024: the three-stage form (kernels_rr.hip) */
    if (d->L <= 32) return fa_launch_r2crows1(d, bufs, tables, cs, cn, st);      /* one butterfly per row (kernels_r1.hip) */
    const bool mixed2 = fa_hip_r2c_rows_tile(d->L) <= 0 && fa_hip_r2c_rows2m_tile(d->L) > 0;
    if (fa_hip_r2c_rows_tile(d->L) <= 0 && !mixed2 && fa_hip_r2c_rows3_tile(d->L) > 0)
        return fa_launch_r2crows3(d, bufs, tables, cs, cn, st);
    R2CRArgs ra;
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    const int T = mixed2 ? fa_hip_r2c_rows2m_tile(d->L) : fa_hip_r2c_rows_tile(d->L);
    const bool fwd = (d->flags & FFTW_AMD_F_R2C_ROWS) != 0;
    const int epi = (int)d->aux_valid;                 /* fused r2r epilogue (r2c) / prologue (c2r), or 0 */
    const int pre = (fwd && d->aux_buf > 0) ? d->aux_buf : 0;   /* in-row gather of the r2r pre-processing */
    const int post = (!fwd && d->aux_buf > 0) ? d->aux_buf : 0; /* output shuffle in the c2r rows store */
    const bool real_src = (!fwd && epi) || pre, real_dst = (fwd && epi) || post;
    if (T <= 0 || d->tile != T || (!real_src && (d->src_im != 1 || d->is_l != 2)) ||
        (!real_dst && (d->dst_im != 1 || d->os_l != 2)) || (mixed2 && (epi || pre || post))) {
        fprintf(stderr, "fftw3_amd: internal error: fused r2c rows step with an unsupported layout\n");
        abort();
    }
    for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) {
        ra.dn[i] = (i < d->ndims) ? d->dim_n[i] : 1;
        ra.dis[i] = (i < d->ndims) ? d->dim_is[i] : 0;
        ra.dos[i] = (i < d->ndims) ? d->dim_os[i] : 0;
    }
    if (bd >= 0) {
        sbase += chunk_adv(d->src_buf, cs, d->dim_is[bd]);
        dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
        ra.dn[bd] = cn;
    }
    ra.src = bufs[d->src_buf] + sbase;
    ra.dst = bufs[d->dst_buf] + dbase;
    ra.os_k = d->os_l;
    ra.dst_im = d->dst_im;
    ra.is_k = d->is_l;
    ra.src_im = d->src_im;
    ra.flags = 0;
    ra.r2r = epi;
    ra.pre = pre;
    ra.post = post;
    ra.twmul = (epi && d->aux_base > 0) ? (int)d->aux_base : 1;
    ra.rn = epi == FFTW_AMD_R2R_POST_E00 ? d->aux_n / 2 + 1 : (epi == FFTW_AMD_R2R_POST_O00 ? d->aux_n / 2 - 1 : d->aux_n);
    if ((!real_src && ((uintptr_t)ra.src % 16)) || (!real_dst && ((uintptr_t)ra.dst % 16))) {
        fprintf(stderr, "fftw3_amd: fftw_execute_dft_r2c needs arrays aligned like the ones the plan was "
                        "created with (16 bytes)\n");
        abort();
    }
    ra.wL = (const cplx *)tables[d->table];
    ra.tw_lo = (const cplx *)tables[d->tw_lo];
    ra.tw_hi = (const cplx *)tables[d->tw_hi];
    ra.tw_shift = d->tw_shift;
    ra.ndims = d->ndims;
    ra.ntiles = (ra.dn[0] + T - 1) / T;
    i64 nblocks = ra.ntiles;
    for (int i = 1; i < d->ndims; ++i) nblocks *= ra.dn[i];
    if (nblocks <= 0) return 0;
    dim3 grid;
    if (nblocks <= 0x7fffffffLL) grid = dim3((unsigned)nblocks, 1, 1);
    else {
        unsigned gy = (unsigned)((nblocks + 0x3fffffffLL) / 0x40000000LL);
        while (nblocks % gy) ++gy;
        grid = dim3((unsigned)(nblocks / gy), gy, 1);
    }
    const bool inverse = (d->flags & FFTW_AMD_F_C2R_ROWS) != 0;
    switch (d->L) {
    case 64: launch_r2cr<8, 8>(ra, grid, st, inverse); break;
    case 128: launch_r2cr<16, 8>(ra, grid, st, inverse); break;
    case 256: launch_r2cr<16, 16>(ra, grid, st, inverse); break;
    case 512: launch_r2cr<32, 16>(ra, grid, st, inverse); break;
    case 1024: launch_r2cr<32, 32>(ra, grid, st, inverse); break;
    default:
        if (fa_launch_r2crows2m(d->L, ra, grid, st, inverse)) {
            fprintf(stderr, "fftw3_amd: internal error: no fused real-rows kernel for half length %d\n", d->L);
            abort();
        }
    }
    return 0;
}
