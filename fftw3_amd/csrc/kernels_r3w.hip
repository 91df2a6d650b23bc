/*
 * kernels_r3w.hip -- instantiations and launcher of the WIDE three-stage rows kernels: contiguous rows of the
 * 7-smooth lengths 8193 ... 16383 of r3w_menu.inc (X(L, R1, R2, R3) entries chosen by tools/gen_r3w_menu.py), one row
 * per workgroup of 512 work-items (pass3g_kernel<R1, R2, R3, 0, 512>, one workgroup per CU like pass3w.hpp's 16384).
 * One trip over HBM where the two-pass plan made two (10 000 = 40 x 250: 38 % of the roofline).  Like the other rows
 * longer than 4096 points these steps have no LDS-kernel fallback (planner: long_rows_ok).
 * Reference counterpart: nested Cooley-Tukey nodes inside one plan (fftw/fftw_api.c:2078-2202).
 */
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "pass3s.hpp"
#include "pass3g.hpp"

template <int R1, int R2, int R3>
static void launch_3gw(const P3SArgs &pa, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    typedef P3GGeom<R1, R2, R3, 512> G;
    static_assert(G::fits && G::T == 1, "wide menu entry: one row per workgroup, at most 32 elements per item and stage");
    const size_t lds = G::lds_doubles * sizeof(double);
    static_assert(G::lds_doubles * sizeof(double) <= 160 * 1024, "wide menu entry exceeds the LDS");
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass3g_kernel<R1, R2, R3, 0, 512>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL((pass3g_kernel<R1, R2, R3, 0, 512>), grid, dim3(512), lds, st, pa);
}

/* 1 when L is a wide menu length */
extern "C" int fa_hip_r3w_has(int L) {
    switch (L) {
#define X(L_, R1_, R2_, R3_) case L_: return 1;
#include "r3w_menu.inc"
#undef X
    }
    return 0;
}

/* contiguous rows of a wide menu length in one pass; 1 = not applicable */
int fa_launch_pass3gw(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                      i64 cs, i64 cn, hipStream_t st) {
    P3SArgs pa = P3SArgs();
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    if (!fa_hip_r3w_has(d->L) || d->tile != 1 || d->src_im != 1 || d->dst_im != 1 || d->tw_n || d->is_l != 2 || d->os_l != 2 ||
        d->tile_lo_n > 1 || (d->flags & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT | FFTW_AMD_F_CONJ_OUT | FFTW_AMD_F_LO_DFT)))
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
    pa.ntiles = pa.dn[0];
    i64 nblocks = pa.ntiles;
    for (int i = 1; i < d->ndims; ++i) nblocks *= pa.dn[i];
    if (nblocks <= 0) return 0;
    if (nblocks > 0x7fffffffLL) return 1;
    dim3 grid((unsigned)nblocks, 1, 1);
    switch (d->L) {
#define X(L_, R1_, R2_, R3_) case L_: launch_3gw<R1_, R2_, R3_>(pa, grid, st); return 0;
#include "r3w_menu.inc"
#undef X
    }
    return 1;
}

/* ---- fused real rows of n = 2L <-> half spectra for the wide half lengths of r3rw_menu.inc (MODE 1 / 2 of
   pass3g_kernel at 512 work-items): real rows of 18 000 ... 30 720 points in ONE trip where round 2 ran a complex
   pass plus an untangle / tangle step */
template <int R1, int R2, int R3>
static void launch_3gw_real(const P3SArgs &pa, dim3 grid, hipStream_t st, bool inverse) {
    static std::atomic<unsigned> attr_done{0};
    typedef P3GGeom<R1, R2, R3, 512> G;
    static_assert(G::fits && G::T == 1, "wide real menu entry: one row per workgroup");
    static_assert(G::lds_doubles * sizeof(double) <= 160 * 1024, "wide real menu entry exceeds the LDS");
    const size_t lds = G::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass3g_kernel<R1, R2, R3, 1, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FA_CHECK(hipFuncSetAttribute((const void *)pass3g_kernel<R1, R2, R3, 2, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    const dim3 g512(512);
    if (inverse) hipLaunchKernelGGL((pass3g_kernel<R1, R2, R3, 2, 512>), grid, g512, lds, st, pa);
    else hipLaunchKernelGGL((pass3g_kernel<R1, R2, R3, 1, 512>), grid, g512, lds, st, pa);
}

extern "C" int fa_hip_r2c_rows3gw_has(int L) {
    switch (L) {
#define X(L_, R1_, R2_, R3_) case L_: return 1;
#include "r3rw_menu.inc"
#undef X
    }
    return 0;
}

/* pa, grid: filled by fa_launch_r2crows3 (kernels_rr.hip); 1 = no kernel for this length */
int fa_launch_r2crows3gw(int L, const P3SArgs &pa, dim3 grid, hipStream_t st, bool inverse) {
    switch (L) {
#define X(L_, R1_, R2_, R3_) case L_: launch_3gw_real<R1_, R2_, R3_>(pa, grid, st, inverse); return 0;
#include "r3rw_menu.inc"
#undef X
    }
    return 1;
}
