/*
 * kernels_bluew.hip -- the one-kernel Bluestein (pass3b.hpp) for padded lengths 8193 ... 16384 (bluew_menu.inc):
 * one row per workgroup of 512 work-items, i.e. lengths n up to 8192 -- every prime below 8192 -- in one kernel
 * instead of the five steps of the step-by-step plan.  A translation unit of its own; kernels_blue.hip dispatches
 * here for padded lengths above 8192.
 */
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "pass3s.hpp"
#include "pass3g.hpp"
#include "pass3b.hpp"

template <int R1, int R2, int R3>
static void launch_bluew(const BlueArgs &ba, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    typedef P3GGeom<R1, R2, R3, 512> G;
    static_assert(G::fits && G::T == 1, "menu entry exceeds the per-item element budget");
    const size_t lds = G::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)blue3g_kernel<R1, R2, R3, 512>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL((blue3g_kernel<R1, R2, R3, 512>), grid, dim3(512), lds, st, ba);
}

/* smallest padded length >= need among the wide kernels (0: none) */
int fa_hip_bluew_nb(int need) {
    static const int nbs[] = {
#define X(L_, R1_, R2_, R3_) L_,
#include "bluew_menu.inc"
#undef X
    };
    for (size_t i = 0; i < sizeof(nbs) / sizeof(nbs[0]); ++i)
        if (nbs[i] >= need) return nbs[i];
    return 0;
}

/* 1 = nb is a wide padded length (one row per tile) */
int fa_hip_bluew_has(int nb) {
    switch (nb) {
#define X(L_, R1_, R2_, R3_) case L_: return 1;
#include "bluew_menu.inc"
#undef X
    }
    return 0;
}

int fa_launch_bluew(int nb, const BlueArgs &ba, dim3 grid, hipStream_t st) {
    switch (nb) {
#define X(L_, R1_, R2_, R3_) case L_: launch_bluew<R1_, R2_, R3_>(ba, grid, st); return 0;
#include "bluew_menu.inc"
#undef X
    }
    return 1;
}
