/*
 * kernels_r3r.hip -- the fused real forms of the general three-stage rows kernel (pass3g_kernel MODE 1 / 2:
 * r2c untangle after stage C, c2r tangle in front of stage A) for the lengths of r3r_menu.inc.  A translation
 * unit of its own: it compiles beside kernels_r3.hip.
 */
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "pass3s.hpp"
#include "pass3g.hpp"

template <int R1, int R2, int R3>
static void launch_3g_real(const P3SArgs &pa, dim3 grid, hipStream_t st, bool inverse) {
    static std::atomic<unsigned> attr_done{0};
    static_assert(P3GGeom<R1, R2, R3>::fits, "menu entry exceeds the per-item element budget");
    const size_t lds = P3GGeom<R1, R2, R3>::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass3g_kernel<R1, R2, R3, 1>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FA_CHECK(hipFuncSetAttribute((const void *)pass3g_kernel<R1, R2, R3, 2>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    if (inverse) hipLaunchKernelGGL((pass3g_kernel<R1, R2, R3, 2>), grid, dim3(256), lds, st, pa);
    else hipLaunchKernelGGL((pass3g_kernel<R1, R2, R3, 1>), grid, dim3(256), lds, st, pa);
}

extern "C" int fa_hip_r2c_rows3gw_has(int L);     /* kernels_r3w.hip: the 512-item forms for half lengths above 8192 */
int fa_launch_r2crows3gw(int L, const P3SArgs &pa, dim3 grid, hipStream_t st, bool inverse);

/* rows per tile of the fused real form for half length L (0: none) */
extern "C" int fa_hip_r2c_rows3g_tile(int L) {
    if (L > 8192 && fa_hip_r2c_rows3gw_has(L)) return 1;
    switch (L) {
#define X(L_, R1_, R2_, R3_) case L_: return P3GGeom<R1_, R2_, R3_>::T;
#include "r3r_menu.inc"
#undef X
    }
    return 0;
}

/* pa, grid: filled by fa_launch_r2crows3 (kernels_rr.hip); 1 = no kernel for this length */
int fa_launch_r2crows3g(int L, const P3SArgs &pa, dim3 grid, hipStream_t st, bool inverse) {
    if (L > 8192) return fa_launch_r2crows3gw(L, pa, grid, st, inverse);
    switch (L) {
#define X(L_, R1_, R2_, R3_) case L_: launch_3g_real<R1_, R2_, R3_>(pa, grid, st, inverse); return 0;
#include "r3r_menu.inc"
#undef X
    }
    return 1;
}
