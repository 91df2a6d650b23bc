// Stand-alone timing harness (design evidence, not product, not a parity test): every ordered radix pair
// (R1, R2) of the short two-stage lengths as a dense-rows pass (L,L,0) and as a column pass (T,T,0), 1 GiB in and
// 1 GiB out each.  The menu generator picks pairs by register spills only; this shows which of the spill-free
// pairs is fastest.  Table contents are zero (timing only).
// Build: hipcc -O3 --offload-arch=gfx950 -I../../include -I../../fftw3_amd/csrc rr_pairs.hip -o rr_pairs
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include <string.h>
#include <algorithm>
#include <vector>

static const i64 NELEM = 1LL << 26;
static double *g_in, *g_out;
static cplx *g_w;
static hipStream_t g_st;

template <int R1, int R2, bool IN_T, bool OUT_T>
static double time_form(int flags = 0) {
    typedef RRGeom<R1, R2> G;
    constexpr int L = R1 * R2, T = G::T;
    P1024Args a;
    memset((void *)&a, 0, sizeof(a));
    for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) a.dn[i] = 1;
    a.src = g_in; a.dst = g_out; a.w1024 = g_w; a.flags = flags;
    if (!IN_T) {                       /* dense rows */
        a.ndims = 1;
        a.dn[0] = NELEM / L; a.dis[0] = a.dos[0] = 2 * L; a.is_l = a.os_l = 2;
    } else {                           /* 4096 interleaved columns per block of L x 4096 */
        a.ndims = 2;
        a.dn[0] = 4096; a.dis[0] = a.dos[0] = 2; a.is_l = a.os_l = 2 * 4096;
        a.dn[1] = NELEM / (4096LL * L); a.dis[1] = a.dos[1] = 2LL * 4096 * L;
    }
    a.ntiles = (a.dn[0] + T - 1) / T;
    i64 nblocks = a.ntiles * a.dn[1];
    const size_t lds = G::lds_doubles * sizeof(double);
    FA_CHECK(hipFuncSetAttribute((const void *)passrr_kernel<R1, R2, IN_T, OUT_T, 0>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    FA_CHECK(hipEventCreate(&e0)); FA_CHECK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 6; ++it) {
        FA_CHECK(hipEventRecord(e0, g_st));
        hipLaunchKernelGGL((passrr_kernel<R1, R2, IN_T, OUT_T, 0>), dim3((unsigned)nblocks), dim3(256), lds, g_st, a);
        FA_CHECK(hipEventRecord(e1, g_st));
        FA_CHECK(hipEventSynchronize(e1));
        float m; FA_CHECK(hipEventElapsedTime(&m, e0, e1));
        if (it) ms.push_back(m);
    }
    FA_CHECK(hipEventDestroy(e0)); FA_CHECK(hipEventDestroy(e1));
    return *std::min_element(ms.begin(), ms.end());
}

template <int R1, int R2> static void run_pair() {
    typedef RRGeom<R1, R2> G;
    const double gb = 32.0 * (NELEM / (R1 * R2)) * (R1 * R2) / 1e9;
    double tr = time_form<R1, R2, false, false>();
    double tc = time_form<R1, R2, true, true>();
    double tri = time_form<R1, R2, false, false>(FFTW_AMD_F_NT_IN), tro = time_form<R1, R2, false, false>(FFTW_AMD_F_NT_OUT);
    double tcb = time_form<R1, R2, true, true>(FFTW_AMD_F_NT_IN | FFTW_AMD_F_NT_OUT);
    printf("L=%-4d %2d x %-2d tile=%-4d wgs=%d  rows %5.2f TB/s (nt-in %5.2f, nt-out %5.2f)   cols %5.2f TB/s (nt %5.2f)\n", R1 * R2, R1, R2, G::T,
           fa_rr_wgs(R1, R2), gb / tr, gb / tri, gb / tro, gb / tc, gb / tcb);
    fflush(stdout);
}

int main() {
    FA_CHECK(hipStreamCreate(&g_st));
    FA_CHECK(hipMalloc(&g_in, NELEM * 16)); FA_CHECK(hipMalloc(&g_out, NELEM * 16));
    FA_CHECK(hipMalloc(&g_w, 4096 * 16));
    FA_CHECK(hipMemset(g_in, 0, NELEM * 16)); FA_CHECK(hipMemset(g_w, 0, 4096 * 16));
#define P(a, b) run_pair<a, b>();
#include "rr_pairs.inc"
#undef P
    return 0;
}
