// Micro-benchmark (design evidence, not product): where the time of one workgroup of the 512-item strided /
// transposed three-stage kernel (pass3t_kernel<8, 16, 16, ., ., 512>, 8 sequences of 2048 points = 256 KiB per tile,
// one workgroup per CU) goes.  Wave 0 of every workgroup stamps the wall clock (100 MHz) at the phase boundaries
// (FA_P3T_STAMP in pass3g.hpp, compiled in only here).  Both passes of the two-trip 2^22 plan on B transforms.
// Build: hipcc -O3 --offload-arch=gfx950 -I../../include -I../../fftw3_amd/csrc -std=c++17 p3t_timeline.hip -o p3t_timeline
#define FA_P3T_TIMELINE 1
#include "common.hpp"
#include "pass1024.hpp"
#include "passrr.hpp"
#include "pass3s.hpp"
#include "pass3g.hpp"
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool IN_T, int TW>
static void run(const char *name, P1024Args pa, int nblocks, long long *dbg_dev, int reps) {
    typedef P3TGeom<8, 16, 16, 512> G;
    const size_t lds = G::lds_doubles * sizeof(double);
    CK(hipFuncSetAttribute((const void *)pass3t_kernel<8, 16, 16, IN_T, TW, 512, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((pass3t_kernel<8, 16, 16, IN_T, TW, 512, 0>), dim3(nblocks), dim3(512), lds, 0, pa);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<long long> h((size_t)nblocks * 16);
    CK(hipMemcpy(h.data(), dbg_dev, h.size() * 8, hipMemcpyDeviceToHost));
    long long tmin = h[0];
    for (int b = 0; b < nblocks; ++b) tmin = std::min(tmin, h[(size_t)b * 16]);
    // phases: 0 start, 7 loads arrived, 1 stage A done, 2 exchange 1 done, 3 stage B done, 4 exchange 2 done, 5 stores issued, 6 stores acked
    const int order[8] = { 0, 7, 1, 2, 3, 4, 5, 6 };
    const char *pn[7] = { "load", "stageA", "exch1", "stageB", "exch2", "stageC+issue", "store-ack" };
    double sum[2][7] = { { 0 } }, start[2] = { 0, 0 }, endt[2] = { 0, 0 };
    int cnt[2] = { 0, 0 };
    for (int b = 0; b < nblocks; ++b) {
        const long long *s = &h[(size_t)b * 16];
        int round = (s[0] - tmin) > 500 ? 1 : 0;                    // started more than 5 us after the first: a later round
        for (int k = 0; k < 7; ++k) sum[round][k] += (double)(s[order[k + 1]] - s[order[k]]) * 0.01;
        start[round] += (double)(s[0] - tmin) * 0.01;
        endt[round] += (double)(s[6] - tmin) * 0.01;
        ++cnt[round];
    }
    printf("%s: %d workgroups, kernel %.1f us (%.2f TB/s for %.0f MiB in + out)\n", name, nblocks, best * 1e3,
           (double)nblocks * 524288.0 / (best * 1e-3) / 1e12, (double)nblocks * 0.5);
    for (int r = 0; r < 2; ++r) {
        if (!cnt[r]) continue;
        printf("  %s (%d wgs): start %.1f us  ", r ? "later rounds" : "first round ", cnt[r], start[r] / cnt[r]);
        for (int k = 0; k < 7; ++k) printf("%s %.1f  ", pn[k], sum[r][k] / cnt[r]);
        printf("end %.1f us\n", endt[r] / cnt[r]);
    }
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2;                     // transforms of 2048 x 2048 complex
    const int stagger = argc > 2 ? atoi(argv[2]) : 0;               // delay of every other first-round workgroup, x 0.64 us
    const i64 N = 2048LL * 2048;
    double *src, *dst;
    cplx *w, *lo, *hi;
    long long *dbg;
    CK(hipMalloc(&src, (size_t)B * N * 16)); CK(hipMalloc(&dst, (size_t)B * N * 16));
    CK(hipMemset(src, 0, (size_t)B * N * 16)); CK(hipMemset(dst, 0, (size_t)B * N * 16));
    std::vector<cplx> hw(2048), hlo(2048), hhi(2048);
    for (int m = 0; m < 2048; ++m) {
        hw[m].x = cos(2 * M_PI * m / 2048.0); hw[m].y = sin(2 * M_PI * m / 2048.0);
        hlo[m].x = cos(2 * M_PI * m / (double)N); hlo[m].y = sin(2 * M_PI * m / (double)N);
        hhi[m].x = cos(2 * M_PI * m / 2048.0); hhi[m].y = sin(2 * M_PI * m / 2048.0);
    }
    CK(hipMalloc(&w, 2048 * 16)); CK(hipMalloc(&lo, 2048 * 16)); CK(hipMalloc(&hi, 2048 * 16));
    CK(hipMemcpy(w, hw.data(), 2048 * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(lo, hlo.data(), 2048 * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(hi, hhi.data(), 2048 * 16, hipMemcpyHostToDevice));
    const int nblocks = 256 * B;
    CK(hipMalloc(&dbg, (size_t)nblocks * 16 * 8));
    P1024Args pa = P1024Args();
    pa.src = src; pa.dst = dst; pa.w1024 = w; pa.tw_lo = lo; pa.tw_hi = hi; pa.tw_shift = 11;
    pa.ndims = 2; pa.flags = 0; pa.dbg = dbg; pa.ntiles = 256; pa.lo_sh = stagger;
    for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) { pa.dn[i] = 1; pa.dis[i] = 0; pa.dos[i] = 0; pa.dtw[i] = 0; }
    // pass 1: columns of the [2048][2048] view, in place order
    pa.is_l = 2 * 2048; pa.os_l = 2 * 2048;
    pa.dn[0] = 2048; pa.dis[0] = 2; pa.dos[0] = 2; pa.dtw[0] = 1;
    pa.dn[1] = B; pa.dis[1] = 2 * N; pa.dos[1] = 2 * N;
    run<true, 0>("columns (IN_T, no twiddle)", pa, nblocks, dbg, 5);
    // pass 2: rows in, transposed store, twiddle on the input
    pa.is_l = 2; pa.os_l = 2 * 2048;
    pa.dn[0] = 2048; pa.dis[0] = 2 * 2048; pa.dos[0] = 2; pa.dtw[0] = 1;
    run<false, 2>("rows in / transposed out, twiddle in", pa, nblocks, dbg, 5);
    return 0;
}
