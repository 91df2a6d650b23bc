// Stand-alone timing harness (design evidence, not product, not a parity test): the two
// pass1024 launches of the N = 2^20 plan, chained over chunks of C transforms exactly as
// fa_run does, with compile-time ablations of the kernel body (pass1024.hpp, ABL) to see
// which part of the non-memory work is exposed.  Results of ablated variants are wrong by
// construction; variant 0 is the product kernel.
// Build: hipcc -O3 --offload-arch=gfx950 -I../../include -I../../fftw3_amd/csrc p1024_ablate.hip -o p1024_ablate
#include "common.hpp"
#include "pass1024.hpp"
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>

static long long *g_dbg[2] = {NULL, NULL};
static void fill_args(P1024Args &a, const double *src, double *dst, bool first, int C,
                      const cplx *w1024, const cplx *lo, const cplx *hi, int nt) {
    memset((void *)&a, 0, sizeof(a));
    for (int i = 0; i < FFTW_AMD_MAX_DIMS; ++i) a.dn[i] = 1;
    a.src = src; a.dst = dst;
    a.ndims = 2;
    a.dn[0] = 1024; a.dn[1] = C;
    a.dis[1] = a.dos[1] = 2097152;
    a.dtw[0] = 1;
    a.os_l = 2048; a.dos[0] = 2;
    if (first) { a.is_l = 2048; a.dis[0] = 2; a.flags = nt ? FFTW_AMD_F_NT_IN : 0; }
    else { a.is_l = 2; a.dis[0] = 2048; a.flags = FFTW_AMD_F_TW_IN | (nt ? FFTW_AMD_F_NT_OUT : 0); }
    a.w1024 = w1024; a.tw_lo = lo; a.tw_hi = hi; a.tw_shift = 10;
    a.ntiles = 128;
    a.dbg = g_dbg[first ? 0 : 1];
}
template <int ABL> static void launch_pair(const double *in, double *scr, double *out, int C, const cplx *w, const cplx *lo,
                                           const cplx *hi, int nt, hipStream_t st) {
    P1024Args a1, a2;
    fill_args(a1, in, scr, true, C, w, lo, hi, nt);
    fill_args(a2, scr, out, false, C, w, lo, hi, nt);
    const size_t lds = FA_P1024_LDS_DOUBLES * sizeof(double);
    static bool done = false;
    if (!done) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass1024_kernel<true, true, 0, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FA_CHECK(hipFuncSetAttribute((const void *)pass1024_kernel<false, true, 2, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done = true;
    }
    hipLaunchKernelGGL((pass1024_kernel<true, true, 0, ABL>), dim3(128 * C), dim3(256), lds, st, a1);
    hipLaunchKernelGGL((pass1024_kernel<false, true, 2, ABL>), dim3(128 * C), dim3(256), lds, st, a2);
}
/* persistent form: exactly 2 workgroups per CU, each walks its XCD's tile range with stride 64 */
template <bool IN_T, bool OUT_T, int HAS_TW>
__global__ void __launch_bounds__(256, 2) pass1024_persist(const P1024Args a, int total) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, per = total >> 3, nslot = gridDim.x >> 3;
    for (int k = slot; k < per; k += nslot) {
        unsigned blk = (unsigned)(x * per + k);
        unsigned tile = blk % (unsigned)a.ntiles, rest = blk / (unsigned)a.ntiles;
        i64 soff = (i64)rest * a.dis[1], doff = (i64)rest * a.dos[1], twb = (i64)rest * a.dtw[1];
        const i64 t0 = (i64)tile * 8;
        P1024Tile t;
        t.lo_sh = 0; t.lo_is = 0; t.lo_os = 0;
        t.src = a.src + soff + t0 * a.dis[0];
        t.dst = a.dst + doff + t0 * a.dos[0];
        t.is_l = a.is_l; t.os_l = a.os_l;
        t.dis0 = a.dis[0]; t.dos0 = a.dos[0];
        t.dtw0 = a.dtw[0]; t.q0 = twb + t0 * a.dtw[0];
        t.w1024 = a.w1024; t.tw_lo = a.tw_lo; t.tw_hi = a.tw_hi; t.tw_shift = a.tw_shift;
        t.Tcur = 8; t.flags = a.flags; t.dbg = NULL;
        p1024_tile<IN_T, OUT_T, HAS_TW, 0>(t, plane, threadIdx.x);
        __syncthreads();
    }
}
/* persistent with DYNAMIC tile fetching: one ticket counter per XCD (blockIdx & 7), the next ticket is fetched
   while the current tile is processed; keeps the XCD-contiguous order and has no static imbalance */
template <bool IN_T, bool OUT_T, int HAS_TW>
__global__ void __launch_bounds__(256, 2) pass1024_dyn(const P1024Args a, int total, unsigned *counters) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    __shared__ unsigned s_ticket;
    const int x = blockIdx.x & 7, per = total >> 3;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&counters[x], 1u);
    __syncthreads();
    unsigned k = s_ticket;
    while (k < (unsigned)per) {
        __syncthreads();
        if (threadIdx.x == 0) s_ticket = atomicAdd(&counters[x], 1u);      /* next ticket: latency hidden by this tile */
        unsigned blk = (unsigned)(x * per) + k;
        unsigned tile = blk % (unsigned)a.ntiles, rest = blk / (unsigned)a.ntiles;
        i64 soff = (i64)rest * a.dis[1], doff = (i64)rest * a.dos[1], twb = (i64)rest * a.dtw[1];
        const i64 t0 = (i64)tile * 8;
        P1024Tile t;
        t.lo_sh = 0; t.lo_is = 0; t.lo_os = 0;
        t.src = a.src + soff + t0 * a.dis[0];
        t.dst = a.dst + doff + t0 * a.dos[0];
        t.is_l = a.is_l; t.os_l = a.os_l;
        t.dis0 = a.dis[0]; t.dos0 = a.dos[0];
        t.dtw0 = a.dtw[0]; t.q0 = twb + t0 * a.dtw[0];
        t.w1024 = a.w1024; t.tw_lo = a.tw_lo; t.tw_hi = a.tw_hi; t.tw_shift = a.tw_shift;
        t.Tcur = 8; t.flags = a.flags; t.dbg = NULL;
        p1024_tile<IN_T, OUT_T, HAS_TW, 0>(t, plane, threadIdx.x);
        __syncthreads();
        k = s_ticket;
    }
}
static unsigned *g_counters = NULL;
static void launch_pair_dyn(const double *in, double *scr, double *out, int C, const cplx *w, const cplx *lo,
                            const cplx *hi, int nt, hipStream_t st) {
    P1024Args a1, a2;
    fill_args(a1, in, scr, true, C, w, lo, hi, nt);
    fill_args(a2, scr, out, false, C, w, lo, hi, nt);
    const size_t lds = FA_P1024_LDS_DOUBLES * sizeof(double);
    static bool done = false;
    if (!done) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass1024_dyn<true, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FA_CHECK(hipFuncSetAttribute((const void *)pass1024_dyn<false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FA_CHECK(hipMalloc(&g_counters, 64));
        done = true;
    }
    FA_CHECK(hipMemsetAsync(g_counters, 0, 64, st));
    hipLaunchKernelGGL((pass1024_dyn<true, true, 0>), dim3(512), dim3(256), lds, st, a1, 128 * C, g_counters);
    hipLaunchKernelGGL((pass1024_dyn<false, true, 2>), dim3(512), dim3(256), lds, st, a2, 128 * C, g_counters + 8);
}
static void launch_pair_persist(const double *in, double *scr, double *out, int C, const cplx *w, const cplx *lo,
                                const cplx *hi, int nt, hipStream_t st) {
    P1024Args a1, a2;
    fill_args(a1, in, scr, true, C, w, lo, hi, nt);
    fill_args(a2, scr, out, false, C, w, lo, hi, nt);
    const size_t lds = FA_P1024_LDS_DOUBLES * sizeof(double);
    static bool done = false;
    if (!done) {
        FA_CHECK(hipFuncSetAttribute((const void *)pass1024_persist<true, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FA_CHECK(hipFuncSetAttribute((const void *)pass1024_persist<false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done = true;
    }
    hipLaunchKernelGGL((pass1024_persist<true, true, 0>), dim3(512), dim3(256), lds, st, a1, 128 * C);
    hipLaunchKernelGGL((pass1024_persist<false, true, 2>), dim3(512), dim3(256), lds, st, a2, 128 * C);
}
/* mixed launch: the tiles of pass 2 of the previous chunk first, then the tiles of pass 1 of this chunk.
   Workgroups are dispatched in order, so pass 1 of chunk k fills the slots that the tail of pass 2 of
   chunk k-1 leaves idle: one dependent launch boundary per chunk instead of two. */
__global__ void __launch_bounds__(256, 2) pass1024_mixed(const P1024Args a2, const P1024Args a1, int n2, int n1) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    const bool second = (int)blockIdx.x < n2;
    const P1024Args &a = second ? a2 : a1;
    const unsigned nb = second ? n2 : n1, b0 = second ? blockIdx.x : blockIdx.x - n2;
    unsigned blk = (nb & 7) ? b0 : (b0 & 7) * (nb >> 3) + (b0 >> 3);
    unsigned tile = blk % (unsigned)a.ntiles, rest = blk / (unsigned)a.ntiles;
    i64 soff = (i64)rest * a.dis[1], doff = (i64)rest * a.dos[1], twb = (i64)rest * a.dtw[1];
    const i64 t0 = (i64)tile * 8;
    P1024Tile t;
    t.lo_sh = 0; t.lo_is = 0; t.lo_os = 0;
    t.src = a.src + soff + t0 * a.dis[0];
    t.dst = a.dst + doff + t0 * a.dos[0];
    t.is_l = a.is_l; t.os_l = a.os_l;
    t.dis0 = a.dis[0]; t.dos0 = a.dos[0];
    t.dtw0 = a.dtw[0]; t.q0 = twb + t0 * a.dtw[0];
    t.w1024 = a.w1024; t.tw_lo = a.tw_lo; t.tw_hi = a.tw_hi; t.tw_shift = a.tw_shift;
    t.Tcur = 8; t.flags = a.flags; t.dbg = NULL;
    if (second) p1024_tile<false, true, 2, 0>(t, plane, threadIdx.x);
    else p1024_tile<true, true, 0, 0>(t, plane, threadIdx.x);
}
static void run_mixed(const double *in, double *scr, double *out, int C, int nchunks, const cplx *w, const cplx *lo,
                      const cplx *hi, hipStream_t st) {
    const i64 N = 1 << 20;
    const size_t lds = FA_P1024_LDS_DOUBLES * sizeof(double);
    static bool done = false;
    if (!done) { FA_CHECK(hipFuncSetAttribute((const void *)pass1024_mixed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    for (int k = 0; k <= nchunks; ++k) {
        P1024Args a1, a2;
        double *slot_w = scr + (i64)(k & 1) * C * N * 2, *slot_r = scr + (i64)((k + 1) & 1) * C * N * 2;
        fill_args(a1, in + (i64)(k < nchunks ? k : 0) * C * N * 2, slot_w, true, C, w, lo, hi, 1);
        fill_args(a2, slot_r, out + (i64)(k > 0 ? k - 1 : 0) * C * N * 2, false, C, w, lo, hi, 1);
        int n1 = k < nchunks ? 128 * C : 0, n2 = k > 0 ? 128 * C : 0;
        hipLaunchKernelGGL(pass1024_mixed, dim3(n1 + n2), dim3(256), lds, st, a2, a1, n2, n1);
    }
}
typedef void (*pairfn)(const double *, double *, double *, int, const cplx *, const cplx *, const cplx *, int, hipStream_t);

int main() {
    const int NT = 256; const i64 N = 1 << 20;
    double *in, *out, *scr; cplx *w, *lo, *hi;
    FA_CHECK(hipMalloc(&in, NT * N * 16)); FA_CHECK(hipMalloc(&out, NT * N * 16)); FA_CHECK(hipMalloc(&scr, (size_t)64 * N * 16));
    FA_CHECK(hipMalloc(&w, 1024 * 16)); FA_CHECK(hipMalloc(&lo, 1024 * 16)); FA_CHECK(hipMalloc(&hi, 1024 * 16));
    {
        std::vector<double> h(NT * N * 2);
        unsigned long long s = 88172645463325252ULL;
        for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
        FA_CHECK(hipMemcpy(in, h.data(), NT * N * 16, hipMemcpyHostToDevice));
        std::vector<double> t(2048);
        for (int k = 0; k < 1024; ++k) { t[2 * k] = cos(2 * M_PI * k / 1024.0); t[2 * k + 1] = sin(2 * M_PI * k / 1024.0); }
        FA_CHECK(hipMemcpy(w, t.data(), 1024 * 16, hipMemcpyHostToDevice));
        for (int k = 0; k < 1024; ++k) { t[2 * k] = cos(2 * M_PI * k / (double)N); t[2 * k + 1] = sin(2 * M_PI * k / (double)N); }
        FA_CHECK(hipMemcpy(lo, t.data(), 1024 * 16, hipMemcpyHostToDevice));
        FA_CHECK(hipMemcpy(hi, w, 1024 * 16, hipMemcpyDeviceToDevice));    // w_N^(1024 k) = w_1024^k
    }
    FA_CHECK(hipMemset(out, 0, NT * N * 16)); FA_CHECK(hipMemset(scr, 0, (size_t)64 * N * 16));
    hipStream_t st; FA_CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; FA_CHECK(hipEventCreate(&e0)); FA_CHECK(hipEventCreate(&e1));
    struct V { const char *name; pairfn f; };
    V vs[] = { {"product kernel", launch_pair<0>}, {"persistent 2 WG/CU", launch_pair_persist}, {"persistent, dynamic tickets", launch_pair_dyn},
               {"product kernel (again)", launch_pair<0>}, {"dynamic tickets (again)", launch_pair_dyn}, {"no butterflies", launch_pair<1>}, {"no twiddles", launch_pair<2>},
               {"no LDS exchange", launch_pair<4>}, {"no bfly, no tw", launch_pair<3>}, {"memory + LDS only... (1|2)", launch_pair<3>},
               {"memory only (1|2|4)", launch_pair<7>} };
    /* warm-up: the first variant timed in a process reads 3-4 % slow (clocks / page tables), which once made
       every later variant look like an improvement */
    for (int r = 0; r < 6; ++r)
        for (int k = 0; k < NT / 16; ++k) launch_pair<0>(in + (i64)k * 16 * N * 2, scr, out + (i64)k * 16 * N * 2, 16, w, lo, hi, 1, st);
    FA_CHECK(hipDeviceSynchronize());
    printf("%-28s %4s %3s | %9s %7s\n", "variant", "C", "nt", "us/xform", "whole%");
    for (int nt : {1}) for (int C : {16, 12, 20, 24, 32}) for (auto &v : vs) {
        if (C != 16 && v.f != (pairfn)launch_pair<0> && v.f != (pairfn)launch_pair_persist && v.f != (pairfn)launch_pair_dyn) continue;
        auto run = [&] { for (int k = 0; k + 1 <= NT / C; ++k) v.f(in + (i64)k * C * N * 2, scr, out + (i64)k * C * N * 2, C, w, lo, hi, nt, st); };
        run(); FA_CHECK(hipDeviceSynchronize());
        double best = 1e30;
        for (int r = 0; r < 3; ++r) {
            FA_CHECK(hipEventRecord(e0, st)); run(); FA_CHECK(hipEventRecord(e1, st)); FA_CHECK(hipEventSynchronize(e1));
            float ms; FA_CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        FA_CHECK(hipGetLastError());
        double us = best * 1e3 / ((NT / C) * C);
        printf("%-28s %4d %3d | %9.2f %7.1f\n", v.name, C, nt, us, 100.0 * 2.0 * N * 16 / us / 1e6 / 8.0);
        fflush(stdout);
    }
    for (int C : {4, 6, 8, 10, 12, 16}) {
        int nch = NT / C;
        run_mixed(in, scr, out, C, nch, w, lo, hi, st); FA_CHECK(hipDeviceSynchronize());
        double best = 1e30;
        for (int r = 0; r < 3; ++r) {
            FA_CHECK(hipEventRecord(e0, st)); run_mixed(in, scr, out, C, nch, w, lo, hi, st); FA_CHECK(hipEventRecord(e1, st)); FA_CHECK(hipEventSynchronize(e1));
            float ms; FA_CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        FA_CHECK(hipGetLastError());
        double us = best * 1e3 / (nch * C);
        printf("%-28s %4d %3d | %9.2f %7.1f\n", "mixed launch (2 slots)", C, 1, us, 100.0 * 2.0 * N * 16 / us / 1e6 / 8.0);
    }
    /* ---- time line of the workgroups of one chunk (ABL = 8: product kernel + time stamps) */
    {
        const int C = 16, nwg = 128 * C;
        FA_CHECK(hipMalloc(&g_dbg[0], nwg * 128)); FA_CHECK(hipMalloc(&g_dbg[1], nwg * 128));
        for (int k = 0; k < 6; ++k) launch_pair<8>(in + (i64)k * C * N * 2, scr, out + (i64)k * C * N * 2, C, w, lo, hi, 1, st);
        FA_CHECK(hipDeviceSynchronize());
        std::vector<long long> h[2];
        for (int p = 0; p < 2; ++p) { h[p].resize(nwg * 16); FA_CHECK(hipMemcpy(h[p].data(), g_dbg[p], nwg * 128, hipMemcpyDeviceToHost)); }
        long long tmin = h[0][0];
        for (int p = 0; p < 2; ++p) for (int i = 0; i < nwg; ++i) if (h[p][i * 16] < tmin) tmin = h[p][i * 16];
        for (int p = 0; p < 2; ++p) {
            std::vector<double> L, Cc, S, D, life, pro, wend;
            long long first = 1LL << 62, last = 0;
            for (int i = 0; i < nwg; ++i) {
                long long *t = &h[p][i * 16];
                L.push_back((t[1] - t[0]) * 0.01); Cc.push_back((t[2] - t[1]) * 0.01); S.push_back((t[3] - t[2]) * 0.01);
                D.push_back((t[4] - t[3]) * 0.01);
                long long e = t[12]; for (int w2 = 1; w2 < 4; ++w2) if (t[12 + w2] > e) e = t[12 + w2];
                pro.push_back((t[0] - t[8]) * 0.01); wend.push_back((e - t[4]) * 0.01); life.push_back((e - t[8]) * 0.01);
                if (t[0] < first) first = t[0];
                if (t[4] > last) last = t[4];
            }
            auto med = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
            printf("pass %d: launch span %.1f us (starts at %+.1f us); per workgroup [us] p10/p50/p90:  load-wait %.1f/%.1f/%.1f  compute+LDS %.1f/%.1f/%.1f  store-issue %.1f/%.1f/%.1f  store-drain %.1f/%.1f/%.1f  life %.1f/%.1f/%.1f  prologue %.2f/%.2f/%.2f  last wave after wave 0 %.1f/%.1f/%.1f\n",
                   p + 1, (last - first) * 0.01, (first - tmin) * 0.01, med(L, .1), med(L, .5), med(L, .9), med(Cc, .1), med(Cc, .5), med(Cc, .9),
                   med(S, .1), med(S, .5), med(S, .9), med(D, .1), med(D, .5), med(D, .9), med(life, .1), med(life, .5), med(life, .9), med(pro, .1), med(pro, .5), med(pro, .9), med(wend, .1), med(wend, .5), med(wend, .9));
            // one CU's time line
            long long key = -1;
            printf("  time line of one CU (us from launch start): start | loads done | stores begin | stores issued | drained  [blk]\n");
            std::vector<std::pair<long long, int>> order;
            for (int i = 0; i < nwg; ++i) {
                long long *t = &h[p][i * 16];
                long long k2 = ((t[6] & 15) << 16) | (t[5] & 0xff00);     // xcc, se, sh, cu
                if (key < 0) key = k2;
                if (k2 == key) order.push_back({t[0], i});
            }
            std::sort(order.begin(), order.end());
            for (auto &o : order) {
                long long *t = &h[p][o.second * 16];
                long long e = t[12]; for (int w2 = 1; w2 < 4; ++w2) if (t[12 + w2] > e) e = t[12 + w2];
                printf("    first instr %7.2f  all waves done %7.2f :", (t[8] - first) * 0.01, (e - first) * 0.01);
                printf("    %7.2f | %7.2f | %7.2f | %7.2f | %7.2f  [%lld]\n", (t[0] - first) * 0.01, (t[1] - first) * 0.01, (t[2] - first) * 0.01,
                       (t[3] - first) * 0.01, (t[4] - first) * 0.01, t[7]);
            }
        }
    }
    return 0;
}
