/*
 * kernels_blue.hip -- instantiations and launcher of the one-kernel Bluestein (pass3b.hpp) for the padded lengths
 * of blue_menu.inc.  A translation unit of its own.
 */
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "pass3s.hpp"
#include "pass3g.hpp"
#include "pass3b.hpp"

template <int R1, int R2, int R3>
static void launch_blue(const BlueArgs &ba, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    static_assert(P3GGeom<R1, R2, R3>::fits, "menu entry exceeds the per-item element budget");
    const size_t lds = P3GGeom<R1, R2, R3>::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)blue3g_kernel<R1, R2, R3>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL((blue3g_kernel<R1, R2, R3>), grid, dim3(256), lds, st, ba);
}

/* kernels_bluew.hip: padded lengths 8193 ... 16384, 512 work-items per row */
int fa_hip_bluew_nb(int need);
int fa_hip_bluew_has(int nb);
int fa_launch_bluew(int nb, const BlueArgs &ba, dim3 grid, hipStream_t st);

/* smallest padded length >= need the kernel is built for (0: none) */
extern "C" int fa_hip_blue_nb(int need) {
    static const int nbs[] = {
#define X(L_, R1_, R2_, R3_) L_,
#include "blue_menu.inc"
#undef X
    };
    for (size_t i = 0; i < sizeof(nbs) / sizeof(nbs[0]); ++i)
        if (nbs[i] >= need) return nbs[i];
    return fa_hip_bluew_nb(need);
}

/* rows per tile for the padded length nb (0: none) */
extern "C" int fa_hip_blue_tile(int nb) {
    switch (nb) {
#define X(L_, R1_, R2_, R3_) case L_: return P3GGeom<R1_, R2_, R3_>::T;
#include "blue_menu.inc"
#undef X
    }
    return fa_hip_bluew_has(nb) ? 1 : 0;
}

/* The step (variant FFTW_AMD_K_BLUE: L = nb, aux_n = n, tw_lo = chirp table, tw_hi = kernel table) has no other
   executor; the planner emits it only for interleaved unit-stride rows of aligned arrays, so anything else here is
   a caller error (new-array execution with differently aligned arrays) and fails loudly. */
int fa_launch_blue(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                   i64 cs, i64 cn, hipStream_t st) {
    BlueArgs ba;
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    const int T = fa_hip_blue_tile(d->L);
    if (T <= 0 || d->tile != T || d->src_im != 1 || d->dst_im != 1 || d->tile_lo_n > 1 || d->tw_n ||
        (d->is_l % 2) || (d->os_l % 2) || d->aux_n < 2 || 2 * d->aux_n - 1 > d->L ||
        (d->flags & ~(FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_SWAP_OUT | FFTW_AMD_F_NT_IN | FFTW_AMD_F_NT_OUT))) {
        fprintf(stderr, "fftw3_amd: internal error: Bluestein rows step with an unsupported layout\n");
        abort();
    }
    for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) {
        ba.dn[i] = (i < d->ndims) ? d->dim_n[i] : 1;
        ba.dis[i] = (i < d->ndims) ? d->dim_is[i] : 0;
        ba.dos[i] = (i < d->ndims) ? d->dim_os[i] : 0;
    }
    if (bd >= 0) {
        sbase += chunk_adv(d->src_buf, cs, d->dim_is[bd]);
        dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
        ba.dn[bd] = cn;
    }
    ba.src = bufs[d->src_buf] + sbase;
    ba.dst = bufs[d->dst_buf] + dbase;
    bool odd = ((uintptr_t)ba.src % 16) || ((uintptr_t)ba.dst % 16);
    for (int i = 0; i < d->ndims; ++i)
        if ((ba.dis[i] % 2) || (ba.dos[i] % 2)) odd = true;
    if (odd) {
        fprintf(stderr, "fftw3_amd: fftw_execute_dft needs arrays aligned like the ones the plan was created with "
                        "(16 bytes)\n");
        abort();
    }
    ba.is_l = d->is_l;
    ba.os_l = d->os_l;
    ba.wL = (const cplx *)tables[d->table];
    ba.chirp = (const cplx *)tables[d->tw_lo];
    ba.kern = (const cplx *)tables[d->tw_hi];
    ba.n = (int)d->aux_n;
    ba.ndims = d->ndims;
    ba.flags = d->flags;
    ba.ntiles = (ba.dn[0] + T - 1) / T;
    i64 nblocks = ba.ntiles;
    for (int i = 1; i < d->ndims; ++i) nblocks *= ba.dn[i];
    if (nblocks <= 0) return 0;
    if (nblocks > 0x7fffffffLL) {
        fprintf(stderr, "fftw3_amd: Bluestein rows step with more than 2^31 tiles\n");
        abort();
    }
    dim3 grid((unsigned)nblocks, 1, 1);
    switch (d->L) {
#define X(L_, R1_, R2_, R3_) case L_: launch_blue<R1_, R2_, R3_>(ba, grid, st); return 0;
#include "blue_menu.inc"
#undef X
    }
    return fa_launch_bluew(d->L, ba, grid, st);
}
