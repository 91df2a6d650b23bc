/*
 * rr_dispatch.hpp -- launch of one two-stage register kernel (passrr.hpp) in the lane mapping / twiddle variant a
 * step asks for.  Shared by the translation units that instantiate the kernels of rr_menu.inc: the menu is cut
 * into three ranges of lengths (kernels_rr.hip, kernels_rr1.hip, kernels_rr2.hip) so that they compile side by
 * side -- one unit with all 200 lengths x 6 variants took six minutes.
 */
#ifndef FA_RR_DISPATCH_HPP
#define FA_RR_DISPATCH_HPP

/* lengths up to FA_RR_CUT1 live in kernels_rr.hip, up to FA_RR_CUT2 in kernels_rr1.hip, the rest in kernels_rr2.hip */
#define FA_RR_CUT1 152
#define FA_RR_CUT2 340

template <int R1, int R2, bool IN_T, bool OUT_T, int TW>
static void launch_rr_variant(const P1024Args &pa, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    const size_t lds = RRGeom<R1, R2>::lds_doubles * sizeof(double);
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)passrr_kernel<R1, R2, IN_T, OUT_T, TW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL((passrr_kernel<R1, R2, IN_T, OUT_T, TW>), grid, dim3(256), lds, st, pa);
}

/* the lane mappings that occur in plans: column passes (T,T) with any twiddle
   mode, the transposing last pass (L,T) without or with input twiddle, and the
   contiguous single pass (L,L).  Anything else falls back to the generic kernel. */
/* Nontemporal accesses pay only when every lane run is made of whole 128-byte lines.  In the rows forms a load run
   is R2 elements and a store run R1 elements of one sequence: short or ragged runs leave each line to several
   instructions, which a temporal access merges in L2 and a nontemporal one does not -- measured on dense rows
   (tests/micro/rr_pairs.hip, profiles/r02_rr_pairs_nt.txt): 16 = 2 x 8 stores 5.55 -> 1.85 TB/s, 36 = 2 x 18
   5.10 -> 1.87, 48 = 4 x 12 5.66 -> 4.24, while 8 x 6 and 16 x 3 gain (5.60 -> 5.71, 5.70 -> 5.93); loads
   18 x 2 5.76 -> 4.07, 4 x 16 5.43 -> 5.70.  So the flags stay only for runs of whole, aligned lines. */
template <int R1, int R2>
static int rr_nt_flags(const P1024Args &pa, bool in_t, bool out_t) {
    int flags = pa.flags;
    bool lines_in = (R2 % 8 == 0) && pa.is_l == 2 && ((uintptr_t)pa.src % 128 == 0);
    bool lines_out = (R1 % 8 == 0) && pa.os_l == 2 && ((uintptr_t)pa.dst % 128 == 0);
    for (int i = 0; i < pa.ndims; ++i) {
        if (pa.dis[i] % 16) lines_in = false;
        if (pa.dos[i] % 16) lines_out = false;
    }
    if (!in_t && !lines_in) flags &= ~FFTW_AMD_F_NT_IN;
    if (!out_t && !lines_out) flags &= ~FFTW_AMD_F_NT_OUT;
    return flags;
}

template <int R1, int R2>
static int dispatch_rr(const P1024Args &pa_in, dim3 grid, hipStream_t st, bool in_t, bool out_t, int tw) {
    P1024Args pa = pa_in;
    pa.flags = rr_nt_flags<R1, R2>(pa_in, in_t, out_t);
    if (in_t && out_t) {
        if (tw == 0) { launch_rr_variant<R1, R2, true, true, 0>(pa, grid, st); return 0; }
        if (tw == 1) { launch_rr_variant<R1, R2, true, true, 1>(pa, grid, st); return 0; }
        launch_rr_variant<R1, R2, true, true, 2>(pa, grid, st);
        return 0;
    }
    if (!in_t && out_t) {
        if (tw == 0) { launch_rr_variant<R1, R2, false, true, 0>(pa, grid, st); return 0; }
        if (tw == 2) { launch_rr_variant<R1, R2, false, true, 2>(pa, grid, st); return 0; }
        return 1;
    }
    if (!in_t && !out_t && tw == 0) { launch_rr_variant<R1, R2, false, false, 0>(pa, grid, st); return 0; }
    return 1;
}

#endif /* FA_RR_DISPATCH_HPP */
