/*
 * kernels_r2cm.hip -- the fused real-rows kernels (r2crows.hpp) for the mixed-radix two-stage lengths of
 * r2cr_menu.inc: real rows of n = 2L = 144 ... 1296 (200, 240, 300, 400, 500, 600, 720, 1000, 1200 ...) to half
 * spectra and back in one trip.  Plain r2c / c2r only: the r2r staging code of the kernels assumes the power-of-two
 * tile of 8192 reals, so the planner never asks these lengths for a fused r2r pre / post step.
 */
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "r2crows.hpp"

template <int R1, int R2>
static void launch_r2cr_m(const R2CRArgs &ra, dim3 grid, hipStream_t st, bool inverse) {
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

/* rows per tile for half length L (0: none) */
extern "C" int fa_hip_r2c_rows2m_tile(int L) {
    switch (L) {
#define X(L_, R1_, R2_) case L_: return R2CRGeom<R1_, R2_>::T;
#include "r2cr_menu.inc"
#undef X
    }
    return 0;
}

/* ra, grid: filled by fa_launch_r2crows (kernels_r3.hip); 1 = no kernel for this length */
int fa_launch_r2crows2m(int L, const R2CRArgs &ra, dim3 grid, hipStream_t st, bool inverse) {
    switch (L) {
#define X(L_, R1_, R2_) case L_: launch_r2cr_m<R1_, R2_>(ra, grid, st, inverse); return 0;
#include "r2cr_menu.inc"
#undef X
    }
    return 1;
}
