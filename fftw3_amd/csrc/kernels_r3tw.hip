/*
 * kernels_r3tw.hip -- the strided / transposed three-stage kernels (pass3t_kernel, pass3g.hpp) with 512 work-items
 * per workgroup for the lengths of r3tw_menu.inc (1080 ... 2048): tiles of 16384 elements hold 8 ... 15 sequences,
 * i.e. 128 ... 240-byte segments on the strided side, where the 256-item tiles of kernels_r3.hip hold 4 ... 7
 * (64 ... 112 bytes, about 3.2 TB/s).  Serves a strided axis of such a length in one trip (2-D / 3-D transforms)
 * and the 2048-point first pass of the two-trip 2^21 plan.  One workgroup per CU.
 */
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "pass3s.hpp"
#include "pass3g.hpp"

template <int R1, int R2, int R3, bool IN_T, int TW, int RD = 0>
static void launch_3tw_variant(const P1024Args &pa, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    typedef P3TGeom<R1, R2, R3, 512> G;
    static_assert(G::T >= 8 && G::QA * R1 <= 40 && G::QB * R2 <= 40 && G::QC * R3 <= 40, "wide strided menu entry");
    static_assert(G::lds_doubles * sizeof(double) <= 160 * 1024, "wide strided menu entry exceeds the LDS");
    const size_t lds = G::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass3t_kernel<R1, R2, R3, IN_T, TW, 512, RD>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL((pass3t_kernel<R1, R2, R3, IN_T, TW, 512, RD>), grid, dim3(512), lds, st, pa);
}

template <int R1, int R2, int R3>
static int dispatch_3tw(const P1024Args &pa, dim3 grid, hipStream_t st, bool in_t, bool out_t, int tw) {
    if (in_t && out_t) {
        if (tw == 0) { launch_3tw_variant<R1, R2, R3, true, 0>(pa, grid, st); return 0; }
        if (tw == 1) { launch_3tw_variant<R1, R2, R3, true, 1>(pa, grid, st); return 0; }
        return 1;
    }
    if (!in_t && out_t) {
        if (tw == 0) { launch_3tw_variant<R1, R2, R3, false, 0>(pa, grid, st); return 0; }
        if (tw == 2) { launch_3tw_variant<R1, R2, R3, false, 2>(pa, grid, st); return 0; }
    }
    return 1;
}

/* 1 = the length has the real-decimated rows form (FFTW_AMD_F_REAL_DEC: last trip of a two-trip r2c) */
extern "C" int fa_hip_r3tw_rdec(int L) { return L == 2048; }

/* sequences per tile of the 512-item strided kernel for length L (0: none) */
extern "C" int fa_hip_r3tw_tile(int L) {
    switch (L) {
#define X(L_, R1_, R2_, R3_) case L_: return P3TGeom<R1_, R2_, R3_, 512>::T;
#include "r3tw_menu.inc"
#undef X
    }
    return 0;
}

static inline dim3 grid_of(i64 nblocks) { return dim3((unsigned)nblocks, 1, 1); }

int fa_launch_pass3tw(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                      i64 cs, i64 cn, hipStream_t st) {
    P1024Args pa;
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    const int T = fa_hip_r3tw_tile(d->L);
    const bool rdec = (d->flags & FFTW_AMD_F_REAL_DEC) != 0;
    if (T <= 0 || d->tile != T || d->src_im != 1 || d->dst_im != 1 || d->tile_lo_n > 1 ||
        (d->flags & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT | FFTW_AMD_F_CONJ_OUT | FFTW_AMD_F_LO_DFT)))
        return 1;
    if (rdec && (!fa_hip_r3tw_rdec(d->L) || (d->flags & (FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_SWAP_OUT)) || !(d->flags & FFTW_AMD_F_TW_IN) ||
                 d->tw_n == 0 || d->dim_tw[0] != 1 || d->dim_n[0] < 2 || d->batch_dim == 0))
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
    pa.dbg = NULL;
    pa.ntiles = (pa.dn[0] + T - 1) / T;
    i64 nblocks = pa.ntiles;
    for (int i = 1; i < d->ndims; ++i) nblocks *= pa.dn[i];
    if (nblocks <= 0) return 0;
    if (nblocks > 0x7fffffffLL) return 1;
    if (pa.dn[0] * 4 < T) return 1;              /* a mostly empty tile: the LDS kernel */
    if (rdec) {
        bool in_t = pa.dn[0] > 1 && iabs64(pa.dis[0]) <= iabs64(pa.is_l);
        bool out_t = pa.dn[0] > 1 && iabs64(pa.dos[0]) <= iabs64(pa.os_l);
        if (in_t || !out_t || d->L != 2048) return 1;
        launch_3tw_variant<8, 16, 16, false, 2, 1>(pa, grid_of(nblocks), st);
        return 0;
    }
    dim3 grid((unsigned)nblocks, 1, 1);
    bool in_t = pa.dn[0] > 1 && iabs64(pa.dis[0]) <= iabs64(pa.is_l);
    bool out_t = pa.dn[0] > 1 && iabs64(pa.dos[0]) <= iabs64(pa.os_l);
    int tw = d->tw_n == 0 ? 0 : ((d->flags & FFTW_AMD_F_TW_IN) ? 2 : 1);
    switch (d->L) {
#define X(L_, R1_, R2_, R3_) case L_: return dispatch_3tw<R1_, R2_, R3_>(pa, grid, st, in_t, out_t, tw);
#include "r3tw_menu.inc"
#undef X
    }
    return 1;
}
