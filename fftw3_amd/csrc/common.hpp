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
#include <atomic>

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

/* Function attributes (dynamic LDS size) are per device: a "done" flag is a bit mask over the
   device ordinals, so that a process that drives several GPUs (sharded plans, one host thread
   per device) sets them on each.  Setting twice is harmless; the bit is published after the
   attribute, so no thread can launch on a device before the attribute is there. */
static inline unsigned fa_dev_bit(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return 1u << (dev & 31);
}
static inline bool fa_attr_needed(const std::atomic<unsigned> &mask) { return !(mask.load(std::memory_order_acquire) & fa_dev_bit()); }
static inline void fa_attr_set(std::atomic<unsigned> &mask) { mask.fetch_or(fa_dev_bit(), std::memory_order_release); }

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

/* 16-byte complex access with an optional nontemporal hint (global_load/store_dwordx4 ... nt).
   The planner asks for it on the streams that touch the caller's arrays once (input of the
   first pass, output of the last), so that the Infinity Cache keeps the scratch image between
   two passes instead (tests/micro/membw3.hip: 12.5 -> 10.0 us per 2^20-point transform). */
typedef double fa_d2 __attribute__((ext_vector_type(2)));
template <bool NT> FA_DEV cplx ld_cplx(const double *p) {
    if (NT) {
        fa_d2 v = __builtin_nontemporal_load(reinterpret_cast<const fa_d2 *>(p));
        return c_make(v.x, v.y);
    }
    return *reinterpret_cast<const cplx *>(p);
}
template <bool NT> FA_DEV void st_cplx(double *p, cplx v) {
    if (NT) {
        fa_d2 w = { v.x, v.y };
        __builtin_nontemporal_store(w, reinterpret_cast<fa_d2 *>(p));
    } else {
        *reinterpret_cast<cplx *>(p) = v;
    }
}

/* linear block id -> work id such that XCD x (= block % 8) walks the contiguous range
   [x * n/8, (x+1) * n/8) in launch order; identity when n is not a multiple of 8 */
FA_DEV i64 fa_xcd_remap(i64 blk, i64 n) {
    if (n & 7) return blk;
    return (blk & 7) * (n >> 3) + (blk >> 3);
}

/* runs of R complex values at a fixed stride, plain or nontemporal (FFTW_AMD_F_NT_IN / NT_OUT):
   the policy is uniform over the launch, so the branch is scalar and each side is straight-line */
template <int R> FA_DEV void ld_run(cplx *x, const double *p, i64 step, bool nt) {
    if (nt) {
#pragma unroll
        for (int i = 0; i < R; ++i) x[i] = ld_cplx<true>(p + i * step);
    } else {
#pragma unroll
        for (int i = 0; i < R; ++i) x[i] = ld_cplx<false>(p + i * step);
    }
}
FA_DEV void st_sel(double *p, cplx v, bool nt) {
    if (nt) st_cplx<true>(p, v); else st_cplx<false>(p, v);
}

/* Block id -> (tile, offsets of the loop dims).  One-dimensional grids (every launch below 2^31
   blocks) take the XCD-contiguous order and 32-bit arithmetic: a 64-bit division is ~130
   instructions here, and this prologue sits in front of every workgroup's first load. */
template <bool TW, class A>
FA_DEV void fa_block_offsets(const A &a, i64 &tile, i64 &soff, i64 &doff, i64 &twb) {
    soff = 0; doff = 0; twb = 0;
    if (gridDim.y == 1) {
        unsigned blk = (unsigned)fa_xcd_remap((i64)blockIdx.x, (i64)gridDim.x);
        const unsigned nt = (unsigned)a.ntiles;
        unsigned rest = blk / nt;
        tile = blk - rest * nt;
        for (int d = 1; d < a.ndims; ++d) {
            const unsigned dn = (unsigned)a.dn[d];
            const unsigned q = rest / dn, idx = rest - q * dn;
            rest = q;
            soff += (i64)idx * a.dis[d];
            doff += (i64)idx * a.dos[d];
            if constexpr (TW) twb += (i64)idx * a.dtw[d];
        }
    } else {
        i64 blk = (i64)blockIdx.x + (i64)blockIdx.y * gridDim.x;
        tile = blk % a.ntiles;
        i64 rest = blk / a.ntiles;
        for (int d = 1; d < a.ndims; ++d) {
            i64 idx = rest % a.dn[d];
            rest /= a.dn[d];
            soff += idx * a.dis[d];
            doff += idx * a.dos[d];
            if constexpr (TW) twb += idx * a.dtw[d];
        }
    }
}

/* w^m from the two-level table: (cos, sin)(2 pi m / n) */
FA_DEV cplx tw2(const cplx *lo, const cplx *hi, int shift, i64 m) {
    cplx a = lo[m & ((1LL << shift) - 1)];
    cplx b = hi[m >> shift];
    return c_mul(a, b);
}

/* the same for WAVE-UNIFORM indices m[0 .. N) (the caller guarantees it, e.g. through readfirstlane): 2 N scalar
   loads through the constant cache and one wait, instead of 2 N vector loads that each fetch one and the same entry
   for 64 lanes */
template <int N> FA_DEV void tw2_uniform(cplx *out, const cplx *lo, const cplx *hi, int shift, const i64 *m) {
    fa_d2 va[N], vb[N];
#pragma unroll
    for (int s = 0; s < N; ++s) {
        const cplx *pl = lo + (m[s] & ((1LL << shift) - 1));
        const cplx *ph = hi + (m[s] >> shift);
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(va[s]) : "s"(pl) : "memory");
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(vb[s]) : "s"(ph) : "memory");
    }
#pragma unroll
    for (int s = 0; s < N; ++s)      /* the first wait covers all; the operands tie every use to a wait */
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(va[s]), "+s"(vb[s]));
#pragma unroll
    for (int s = 0; s < N; ++s) out[s] = c_mul(c_make(va[s].x, va[s].y), c_make(vb[s].x, vb[s].y));
}

static inline i64 iabs64(i64 v) { return v < 0 ? -v : v; }

/* offsets applied for the current batch chunk: user buffers advance, scratch
   buffers are reused per chunk */
static inline i64 chunk_adv(int buf, i64 chunk_start, i64 stride) {
    return (buf < 2) ? chunk_start * stride : 0;
}

#endif /* FA_COMMON_HPP */
