/*
 * kernels_rr1.hip -- the two-stage register kernels of rr_menu.inc with FA_RR_CUT1 < L <= FA_RR_CUT2 (see rr_dispatch.hpp).
 */
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "rr_dispatch.hpp"

template <int L_, int R1_, int R2_>
static int part_case(const P1024Args &pa, dim3 grid, hipStream_t st, bool in_t, bool out_t, int tw) {
    if constexpr ((L_ > (FA_RR_CUT1)) && (L_ <= (FA_RR_CUT2))) return dispatch_rr<R1_, R2_>(pa, grid, st, in_t, out_t, tw);
    else return 1;
}

int fa_dispatch_rr_part1(int L, const P1024Args &pa, dim3 grid, hipStream_t st, bool in_t, bool out_t, int tw) {
    switch (L) {
#define X(L_, R1_, R2_) case L_: return part_case<L_, R1_, R2_>(pa, grid, st, in_t, out_t, tw);
#include "rr_menu.inc"
#undef X
    }
    return 1;
}
