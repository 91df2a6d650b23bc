/*
 * kernels_rr.hip -- instantiations and launcher of the two-stage register
 * kernels (passrr.hpp) for every length of rr_menu.inc: the powers of two 64 ... 512
 * and the mixed-radix lengths whose (R1, R2) split tools/gen_rr_menu.py found to
 * compile without register spills (X(L, R1, R2) entries; the reference's counterpart
 * is its codelet list, fftw/dft_scalar/codelets/codlist.c).  A translation unit of its
 * own so that it compiles in parallel with kernels.hip.
 */
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "pass3s.hpp"
#include "pass3w.hpp"

#include "rr_dispatch.hpp"

/* kernels_rr1.hip, kernels_rr2.hip: the upper two thirds of the menu */
int fa_dispatch_rr_part1(int L, const P1024Args &pa, dim3 grid, hipStream_t st, bool in_t, bool out_t, int tw);
int fa_dispatch_rr_part2(int L, const P1024Args &pa, dim3 grid, hipStream_t st, bool in_t, bool out_t, int tw);

template <int L_, int R1_, int R2_>
static int rr_case(const P1024Args &pa, dim3 grid, hipStream_t st, bool in_t, bool out_t, int tw) {
    if constexpr (L_ <= FA_RR_CUT1) return dispatch_rr<R1_, R2_>(pa, grid, st, in_t, out_t, tw);
    else if constexpr (L_ <= FA_RR_CUT2) return fa_dispatch_rr_part1(L_, pa, grid, st, in_t, out_t, tw);
    else return fa_dispatch_rr_part2(L_, pa, grid, st, in_t, out_t, tw);
}

int fa_launch_passrr(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                     i64 cs, i64 cn, hipStream_t st) {
    P1024Args pa;
    int bd = d->batch_dim, T;
    i64 sbase = d->src_base, dbase = d->dst_base;
    if (d->src_im != 1 || d->dst_im != 1 ||
        (d->flags & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT | FFTW_AMD_F_CONJ_OUT)))
        return 1;
    T = fa_hip_rr_tile(d->L);
    if (T <= 0) return 1;
    if (d->tile != T) return 1;                  /* the planner sized the step for another kernel */
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
    if (d->tile_lo_n > 1) {
        if (d->tile_lo_n != 2 && d->tile_lo_n != 4) return 1;
        pa.lo_sh = d->tile_lo_n == 2 ? 1 : 2;
        if ((pa.lo_is % 2) || (pa.lo_os % 2)) return 1;
    }
    if (pa.lo_sh && (T & (T - 1))) return 1;     /* pair tiles need a power-of-two tile */
    T >>= pa.lo_sh;                      /* hi entries per tile */
    pa.ntiles = (pa.dn[0] + T - 1) / T;
    i64 nblocks = pa.ntiles;
    for (int i = 1; i < d->ndims; ++i) nblocks *= pa.dn[i];
    if (nblocks <= 0) return 0;
    if (nblocks > 0x7fffffffLL) return 1;
    /* a tile that would be mostly empty is better served by the generic kernel */
    if (pa.dn[0] * 4 < T) return 1;
    dim3 grid((unsigned)nblocks, 1, 1);
    bool in_t = pa.dn[0] > 1 && iabs64(pa.dis[0]) <= iabs64(pa.is_l);
    bool out_t = pa.dn[0] > 1 && iabs64(pa.dos[0]) <= iabs64(pa.os_l);
    int tw = d->tw_n == 0 ? 0 : ((d->flags & FFTW_AMD_F_TW_IN) ? 2 : 1);
    switch (d->L) {
#define X(L_, R1_, R2_) case L_: return rr_case<L_, R1_, R2_>(pa, grid, st, in_t, out_t, tw);
#include "rr_menu.inc"
#undef X
    }
    return 1;
}

template <int R1>
static void launch_3s(const P3SArgs &pa, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    const size_t lds = P3SGeom<R1>::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass3s_kernel<R1>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL((pass3s_kernel<R1>), grid, dim3(256), lds, st, pa);
}

/* FFTW_AMD_F_LO_DFT steps (rows + a DFT across the rows of a tile): kernels_sq.hip */
int fa_launch_lo_dft(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables, i64 cs, i64 cn, hipStream_t st);

/* contiguous rows of 2048 / 4096 / 8192 / 16384 in one pass; 1 = not applicable */
int fa_launch_pass3s(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                     i64 cs, i64 cn, hipStream_t st) {
    P3SArgs pa;
    int bd = d->batch_dim, T;
    i64 sbase = d->src_base, dbase = d->dst_base;
    if (d->flags & FFTW_AMD_F_LO_DFT) return fa_launch_lo_dft(d, bufs, tables, cs, cn, st);
    if ((d->L != 2048 && d->L != 4096 && d->L != 8192 && d->L != 16384) || d->src_im != 1 || d->dst_im != 1 || d->tw_n ||
        d->is_l != 2 || d->os_l != 2 || d->tile_lo_n > 1 ||
        (d->flags & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT | FFTW_AMD_F_CONJ_OUT)))
        return 1;
    T = d->L == 16384 ? 1 : 8192 / d->L;
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
    if (d->L == 2048) launch_3s<8>(pa, grid, st);
    else if (d->L == 4096) launch_3s<16>(pa, grid, st);
    else if (d->L == 8192) launch_3s<32>(pa, grid, st);
    else {
        /* 16384: one workgroup of 512 items per row (pass3w.hpp); the store flags are template parameters */
        static std::atomic<unsigned> attr_done{0};
        const size_t lds = P3WGeom::lds_doubles * sizeof(double);
        if (fa_attr_needed(attr_done)) {
            FA_CHECK(hipFuncSetAttribute((const void *)pass3w_kernel<0, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            FA_CHECK(hipFuncSetAttribute((const void *)pass3w_kernel<0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            FA_CHECK(hipFuncSetAttribute((const void *)pass3w_kernel<0, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            FA_CHECK(hipFuncSetAttribute((const void *)pass3w_kernel<0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            fa_attr_set(attr_done);
        }
        const bool sw = (d->flags & FFTW_AMD_F_SWAP_OUT) != 0, nt = (d->flags & FFTW_AMD_F_NT_OUT) != 0;
        if (sw && nt) hipLaunchKernelGGL((pass3w_kernel<0, true, true>), grid, dim3(512), lds, st, pa);
        else if (sw) hipLaunchKernelGGL((pass3w_kernel<0, true, false>), grid, dim3(512), lds, st, pa);
        else if (nt) hipLaunchKernelGGL((pass3w_kernel<0, false, true>), grid, dim3(512), lds, st, pa);
        else hipLaunchKernelGGL((pass3w_kernel<0, false, false>), grid, dim3(512), lds, st, pa);
    }
    return 0;
}

/* rows per tile of the fused real-rows form of the three-stage kernel (half length L; plain r2c / c2r without
   r2r hooks), 0: none */
extern "C" int fa_hip_r2c_rows3g_tile(int L);     /* kernels_r3r.hip: the mixed-radix lengths of r3r_menu.inc */
int fa_launch_r2crows3g(int L, const P3SArgs &pa, dim3 grid, hipStream_t st, bool inverse);
extern "C" int fa_hip_r2c_rows3_tile(int L) {
    if (L == 2048 || L == 4096 || L == 8192) return 8192 / L;
    if (L == 16384) return 1;
    return fa_hip_r2c_rows3g_tile(L);
}

template <int R1>
static void launch_3s_real(const P3SArgs &pa, dim3 grid, hipStream_t st, bool inverse) {
    static std::atomic<unsigned> attr_done{0};
    const size_t lds = P3SGeom<R1>::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass3s_kernel<R1, 1>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FA_CHECK(hipFuncSetAttribute((const void *)pass3s_kernel<R1, 2>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    if (inverse) hipLaunchKernelGGL((pass3s_kernel<R1, 2>), grid, dim3(256), lds, st, pa);
    else hipLaunchKernelGGL((pass3s_kernel<R1, 1>), grid, dim3(256), lds, st, pa);
}

/* real rows of n = 2L <-> half spectra in one trip, L = 2048 / 4096 / 8192 (pass3s_kernel MODE 1 / 2) and the
   mixed-radix lengths of r3r_menu.inc (pass3g_kernel MODE 1 / 2, kernels_r3r.hip).  Like the
   two-stage form (fa_launch_r2crows) the step has no other executor: a layout the planner did not promise is a
   caller error (new-array execution with differently aligned arrays) and fails loudly. */
int fa_launch_r2crows3(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                       i64 cs, i64 cn, hipStream_t st) {
    P3SArgs pa;
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    const int T = fa_hip_r2c_rows3_tile(d->L);
    const bool inverse = (d->flags & FFTW_AMD_F_C2R_ROWS) != 0;
    if (T <= 0 || d->tile != T || d->src_im != 1 || d->dst_im != 1 || d->is_l != 2 || d->os_l != 2 ||
        d->aux_valid || d->aux_buf > 0 || d->tile_lo_n > 1) {
        fprintf(stderr, "fftw3_amd: internal error: fused r2c rows step with an unsupported layout\n");
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
    pa.src = bufs[d->src_buf] + sbase;
    pa.dst = bufs[d->dst_buf] + dbase;
    bool odd = ((uintptr_t)pa.src % 16) || ((uintptr_t)pa.dst % 16);
    for (int i = 0; i < d->ndims; ++i)
        if ((pa.dis[i] % 2) || (pa.dos[i] % 2)) odd = true;
    if (odd) {
        fprintf(stderr, "fftw3_amd: fftw_execute_dft_r2c / _c2r needs arrays aligned like the ones the plan was "
                        "created with (16 bytes)\n");
        abort();
    }
    pa.wL = (const cplx *)tables[d->table];
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
        fprintf(stderr, "fftw3_amd: fused r2c rows step with more than 2^31 tiles\n");
        abort();
    }
    dim3 grid((unsigned)nblocks, 1, 1);
    if (d->L == 2048) launch_3s_real<8>(pa, grid, st, inverse);
    else if (d->L == 4096) launch_3s_real<16>(pa, grid, st, inverse);
    else if (d->L == 8192) launch_3s_real<32>(pa, grid, st, inverse);
    else if (d->L == 16384) {
        static std::atomic<unsigned> attr_done{0};
        const size_t lds = P3WGeom::lds_doubles * sizeof(double);
        if (fa_attr_needed(attr_done)) {
            FA_CHECK(hipFuncSetAttribute((const void *)pass3w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            FA_CHECK(hipFuncSetAttribute((const void *)pass3w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            fa_attr_set(attr_done);
        }
        if (inverse) hipLaunchKernelGGL(pass3w_kernel<2>, grid, dim3(512), lds, st, pa);
        else hipLaunchKernelGGL(pass3w_kernel<1>, grid, dim3(512), lds, st, pa);
    }
    else if (fa_launch_r2crows3g(d->L, pa, grid, st, inverse)) {
        fprintf(stderr, "fftw3_amd: internal error: no fused real-rows kernel for half length %d\n", d->L);
        abort();
    }
    return 0;
}

/* tile width the register kernels use for a sub-transform length (0: none) */
extern "C" int fa_hip_rr_tile(int L) {
    switch (L) {
#define X(L_, R1_, R2_) case L_: return RRGeom<R1_, R2_>::T;
#include "rr_menu.inc"
#undef X
    }
    return 0;
}
