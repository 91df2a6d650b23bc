/*
 * r2r_epi.hpp -- the r2r epilogue / prologue hooks shared by the untangle / tangle kernels
 * (kernels.hip) and the fused real-rows kernels (r2crows.hpp).  The argument struct A
 * provides: dst / src, os_k / is_k, dst_im / src_im, flags, r2r (FFTW_AMD_R2R_* or 0),
 * rn (r2r length), tw_lo / tw_hi / tw_shift (modulus 4n for the 10 / 01 kinds).
 */
#ifndef FA_R2R_EPI_HPP
#define FA_R2R_EPI_HPP

/* Fused r2r epilogue of the r2c untangle kernels: Y = half-spectrum entry idx of
   the inner real DFT; what POST_* of r2r_kernel would do with it, without the
   trip through memory.  D = destination row (reals of stride os_k). */
template <class A>
FA_DEV void epi_store(const A &a, i64 doff, i64 idx, cplx Y) {
    if (a.r2r == 0) {
        store_elem<false>(a.dst, doff + idx * a.os_k, a.dst_im, a.flags, Y);
        return;
    }
    double *D = a.dst + doff;
    const i64 n = a.rn;
    const bool mid = (idx > 0 && 2 * idx < n);
    switch (a.r2r) {
    case FFTW_AMD_R2R_POST_R2HC:
        D[idx * a.os_k] = Y.x;
        if (mid) D[(n - idx) * a.os_k] = Y.y;
        break;
    case FFTW_AMD_R2R_POST_DHT:
        if (mid) { D[idx * a.os_k] = Y.x - Y.y; D[(n - idx) * a.os_k] = Y.x + Y.y; }
        else D[idx * a.os_k] = Y.x;
        break;
    case FFTW_AMD_R2R_POST_E10:
    case FFTW_AMD_R2R_POST_O10: {
        const bool rev = (a.r2r == FFTW_AMD_R2R_POST_O10);
        cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, idx);
        double vi = mid ? Y.y : 0.0;
        double dr = Y.x * w.x + vi * w.y, di = vi * w.x - Y.x * w.y;
        D[(rev ? n - 1 - idx : idx) * a.os_k] = 2.0 * dr;
        if (mid) D[(rev ? idx - 1 : n - idx) * a.os_k] = -2.0 * di;
        break;
    }
    case FFTW_AMD_R2R_POST_E00:
        D[idx * a.os_k] = Y.x;
        break;
    case FFTW_AMD_R2R_POST_O00:
        if (idx >= 1 && idx <= n) D[(idx - 1) * a.os_k] = -Y.y;
        break;
    default:
        break;
    }
}

/* Fused r2r prologue of the c2r tangle kernels: half-spectrum entry idx built
   from the user's real r2r input S (stride is_k), as PRE_* of r2r_kernel would. */
template <class A>
FA_DEV cplx pro_load(const A &a, i64 soff, i64 idx) {
    if (a.r2r == 0) return load_elem<false>(a.src, soff + idx * a.is_k, a.src_im, 0);
    const double *S = a.src + soff;
    const i64 n = a.rn;
    switch (a.r2r) {
    case FFTW_AMD_R2R_PRE_HC2R:
        return c_make(S[idx * a.is_k], (idx > 0 && 2 * idx < n) ? S[(n - idx) * a.is_k] : 0.0);
    case FFTW_AMD_R2R_PRE_E01:
    case FFTW_AMD_R2R_PRE_O01: {
        double x, y;
        if (a.r2r == FFTW_AMD_R2R_PRE_E01) { x = S[idx * a.is_k]; y = (idx > 0) ? S[(n - idx) * a.is_k] : 0.0; }
        else { x = S[(n - 1 - idx) * a.is_k]; y = (idx > 0) ? S[(idx - 1) * a.is_k] : 0.0; }
        cplx w = tw2(a.tw_lo, a.tw_hi, a.tw_shift, idx);
        return c_make(x * w.x + y * w.y, x * w.y - y * w.x);
    }
    default:
        return c_make(0.0, 0.0);
    }
}

#endif /* FA_R2R_EPI_HPP */
