/*
 * kernels_sq.hip -- FFTW_AMD_F_LO_DFT steps: the last trip of a two-dimensional transform whose strided axis was
 * split as T x L0 (planner.c emit_rows_lo_dft).  A tile is T whole rows; the kernel transforms the rows and takes the
 * DFT of length T across them in the same registers:
 *     rows of 2048, T = 4   pass3s_kernel<8, 0, true>     256 items, two workgroups per CU
 *     rows of 4096, T = 2   pass3s_kernel<16, 0, true>
 *     rows of 4096, T = 4   pass3q_kernel                 512 items, one workgroup per CU (pass3q.hpp)
 * These steps have no other executor: the planner settles the layout at plan time, and a layout the kernels cannot
 * take is an internal error that fails loudly.
 */
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "pass3s.hpp"
#include "pass3q.hpp"

template <class K>
static void launch_sq(K kernel, std::atomic<unsigned> &attr_done, size_t lds, unsigned nblocks, unsigned nthreads,
                      hipStream_t st, const P3SArgs &pa) {
    if (fa_attr_needed(attr_done)) {
        FA_CHECK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fa_attr_set(attr_done);
    }
    hipLaunchKernelGGL(kernel, dim3(nblocks, 1, 1), dim3(nthreads), lds, st, pa);
}

int fa_launch_lo_dft(const fftw_amd_step_desc *d, double *const *bufs, void *const *tables,
                     i64 cs, i64 cn, hipStream_t st) {
    P3SArgs pa = P3SArgs();
    int bd = d->batch_dim;
    i64 sbase = d->src_base, dbase = d->dst_base;
    const int T = d->tile_lo_n;
    bool bad = !((d->L == 2048 && T == 4) || (d->L == 4096 && (T == 2 || T == 4))) || d->src_im != 1 || d->dst_im != 1 || d->tw_n ||
               d->is_l != 2 || d->os_l != 2 || (d->flags & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT | FFTW_AMD_F_CONJ_OUT)) ||
               (d->tile_lo_is % 2) || (d->tile_lo_os % 2);
    for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) {
        pa.dn[i] = (i < d->ndims) ? d->dim_n[i] : 1;
        pa.dis[i] = (i < d->ndims) ? d->dim_is[i] : 0;
        pa.dos[i] = (i < d->ndims) ? d->dim_os[i] : 0;
        if (i < d->ndims && ((pa.dis[i] % 2) || (pa.dos[i] % 2))) bad = true;
    }
    if (bd >= 0) {
        sbase += chunk_adv(d->src_buf, cs, d->dim_is[bd]);
        dbase += chunk_adv(d->dst_buf, cs, d->dim_os[bd]);
        pa.dn[bd] = cn;
    }
    pa.src = bufs[d->src_buf] + sbase;
    pa.dst = bufs[d->dst_buf] + dbase;
    if (((uintptr_t)pa.src % 16) || ((uintptr_t)pa.dst % 16)) bad = true;
    i64 nblocks = pa.dn[0];
    for (int i = 1; i < d->ndims; ++i) nblocks *= pa.dn[i];
    if (nblocks > 0x7fffffffLL) bad = true;
    if (bad) {
        fprintf(stderr, "fftw3_amd: internal error: rows step with a DFT across the rows of a tile (L = %d, T = %d) in an unsupported layout\n", d->L, T);
        abort();
    }
    if (nblocks <= 0) return 0;
    pa.wL = (const cplx *)tables[d->table];
    pa.ndims = d->ndims;
    pa.flags = d->flags;
    pa.ntiles = pa.dn[0];
    pa.srs = d->tile_lo_is;
    pa.drs = d->tile_lo_os;
    const int outf = ((d->flags & FFTW_AMD_F_SWAP_OUT) ? 1 : 0) | ((d->flags & FFTW_AMD_F_NT_OUT) ? 2 : 0);
    static std::atomic<unsigned> a8[4], a16[4];
    const size_t l8 = P3SGeom<8>::lds_doubles * sizeof(double), l16 = P3SGeom<16>::lds_doubles * sizeof(double);
#define FA_SQ_CASE(F) case F: if (d->L == 2048) launch_sq(pass3s_kernel<8, 0, true, F>, a8[F], l8, (unsigned)nblocks, 256, st, pa); \
                              else launch_sq(pass3s_kernel<16, 0, true, F>, a16[F], l16, (unsigned)nblocks, 256, st, pa); break;
    if (d->L == 2048 || T == 2) {
        switch (outf) { FA_SQ_CASE(0) FA_SQ_CASE(1) FA_SQ_CASE(2) FA_SQ_CASE(3) }
    }
#undef FA_SQ_CASE
    else {
        static std::atomic<unsigned> q00{0}, q01{0}, q10{0}, q11{0};
        const size_t lq = P3QGeom::lds_doubles * sizeof(double);
        const bool sw = (d->flags & FFTW_AMD_F_SWAP_OUT) != 0, nt = (d->flags & FFTW_AMD_F_NT_OUT) != 0;
        if (sw && nt) launch_sq(pass3q_kernel<true, true>, q11, lq, (unsigned)nblocks, 512, st, pa);
        else if (sw) launch_sq(pass3q_kernel<true, false>, q10, lq, (unsigned)nblocks, 512, st, pa);
        else if (nt) launch_sq(pass3q_kernel<false, true>, q01, lq, (unsigned)nblocks, 512, st, pa);
        else launch_sq(pass3q_kernel<false, false>, q00, lq, (unsigned)nblocks, 512, st, pa);
    }
    return 0;
}
