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

template <int R1, int R2, int R3>
static void launch_3g(const P3SArgs &pa, dim3 grid, hipStream_t st) {
    static bool attr_done = false;
    static_assert(P3GGeom<R1, R2, R3>::fits, "menu entry exceeds the per-item element budget");
    const size_t lds = P3GGeom<R1, R2, R3>::lds_doubles * sizeof(double);
    if (!attr_done) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass3g_kernel<R1, R2, R3>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((pass3g_kernel<R1, R2, R3>), grid, dim3(256), lds, st, pa);
}

/* rows per tile of the three-stage kernel for length L (0: none).  2048 / 4096 are the
   tuned pass3s kernels of kernels_rr.hip. */
extern "C" int fa_hip_r3_tile(int L) {
    if (L == 2048 || L == 4096) return 8192 / L;
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
    if (T <= 0 || d->L == 2048 || d->L == 4096 || d->tile != T || d->src_im != 1 || d->dst_im != 1 ||
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
