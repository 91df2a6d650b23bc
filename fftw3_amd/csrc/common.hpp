/*
 * common.hpp -- shared prelude of the HIP translation units (kernels.hip,
 * kernels_rr.hip): element type, access helpers, two-level twiddle lookup.
 */
#ifndef FA_COMMON_HPP
#define FA_COMMON_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define FA_DEV __device__ __forceinline__
#include "butterflies.h"
#include "fa_hip.h"

typedef long long i64;

#define FA_CHECK(call)                                                              \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "fftw3_amd: HIP error %s at %s:%d (%s)\n",             \
                    hipGetErrorString(e_), __FILE__, __LINE__, #call);              \
            abort();                                                                \
        }                                                                           \
    } while (0)

/* ------------------------------------------------------------------------ */
/* element access helpers                                                    */
/* ------------------------------------------------------------------------ */

template <bool VEC>
FA_DEV cplx load_elem(const double *p, i64 a, i64 im, int flags) {
    cplx v;
    if (VEC) {
        v = *reinterpret_cast<const cplx *>(p + a);
    } else {
        v.x = p[a];
        v.y = (flags & FFTW_AMD_F_REAL_IN) ? 0.0 : p[a + im];
    }
    if (flags & FFTW_AMD_F_SWAP_IN) { double t = v.x; v.x = v.y; v.y = t; }
    return v;
}

template <bool VEC>
FA_DEV void store_elem(double *p, i64 a, i64 im, int flags, cplx v) {
    if (flags & FFTW_AMD_F_CONJ_OUT) v.y = -v.y;
    if (flags & FFTW_AMD_F_SWAP_OUT) { double t = v.x; v.x = v.y; v.y = t; }
    if (VEC) {
        *reinterpret_cast<cplx *>(p + a) = v;
    } else {
        p[a] = v.x;
        if (!(flags & FFTW_AMD_F_REAL_OUT)) p[a + im] = v.y;
    }
}

/* w^m from the two-level table: (cos, sin)(2 pi m / n) */
FA_DEV cplx tw2(const cplx *lo, const cplx *hi, int shift, i64 m) {
    cplx a = lo[m & ((1LL << shift) - 1)];
    cplx b = hi[m >> shift];
    return c_mul(a, b);
}

static inline i64 iabs64(i64 v) { return v < 0 ? -v : v; }

/* offsets applied for the current batch chunk: user buffers advance, scratch
   buffers are reused per chunk */
static inline i64 chunk_adv(int buf, i64 chunk_start, i64 stride) {
    return (buf < 2) ? chunk_start * stride : 0;
}

#endif /* FA_COMMON_HPP */
