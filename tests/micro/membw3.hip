// Micro-benchmark (design evidence, not product): the two-pass 2^20-point plan as pure data
// movement -- K1 = column-tile copy input -> scratch (128-B segments at a 16 KiB pitch on both
// sides), K2 = row-tile copy scratch -> output (contiguous 128 KiB reads, 128-B segment stores)
// -- over chunks of C transforms, to find what chunk size / cache policy / stream layout the
// memory system rewards, with no FFT arithmetic in the way.  Also: granularity of the
// tile -> XCD mapping.   Build: hipcc -O3 --offload-arch=gfx950 membw3.hip -o membw3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double2 cplx;
typedef long long i64;

// POL: 0 plain, 1 nontemporal
template <int POL> __device__ __forceinline__ cplx ld(const cplx *p) {
    if (POL == 1) { cplx v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
    return *p;
}
template <int POL> __device__ __forceinline__ void st(cplx *p, cplx v) {
    if (POL == 1) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
    else *p = v;
}
// tile -> XCD mapping: groups of G consecutive tiles go to one XCD (G = 0: identity; G < 0: each XCD
// owns one contiguous eighth of the launch)
__device__ __forceinline__ i64 tile_of(i64 blk, i64 ntiles, int G) {
    if (G == 0) return blk;
    i64 x = blk & 7, j = blk >> 3;
    if (G < 0) return x * (ntiles >> 3) + j;
    return ((j / G) * 8 + x) * G + (j % G);
}
// K1: tile = 1024 rows x 8 columns of a [1024][1024] matrix per transform, both sides
template <int LP, int SP>
__global__ void __launch_bounds__(256, 2) k_col(const cplx *__restrict__ s, cplx *__restrict__ d, i64 ntiles, int G) {
    i64 t = tile_of(blockIdx.x, ntiles, G);
    i64 org = (t >> 7) * (1 << 20) + (t & 127) * 8;
    const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
    const cplx *sp = s + org + c + (i64)r * 1024;
    cplx *dp = d + org + c + (i64)r * 1024;
    cplx v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = ld<LP>(sp + (i64)i * 32 * 1024);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; ++i) st<SP>(dp + (i64)i * 32 * 1024, v[i]);
}
// K2: tile = 8 contiguous rows (8192 elements) in, transposed 1024 x 8 out
template <int LP, int SP>
__global__ void __launch_bounds__(256, 2) k_row(const cplx *__restrict__ s, cplx *__restrict__ d, i64 ntiles, int G) {
    i64 t = tile_of(blockIdx.x, ntiles, G);
    const cplx *sp = s + t * 8192 + threadIdx.x;
    i64 org = (t >> 7) * (1 << 20) + (t & 127) * 8;
    const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
    cplx *dp = d + org + c + (i64)r * 1024;
    cplx v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = ld<LP>(sp + i * 256);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; ++i) st<SP>(dp + (i64)i * 32 * 1024, v[i]);
}

static hipEvent_t e0, e1;
template <class F> static double bestms(F f, int reps = 3) {
    f(); CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}
typedef void (*kfn)(const cplx *, cplx *, i64, int);

int main(int argc, char **argv) {
    const char *only = argc > 1 ? argv[1] : "";
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NT = 256;                          // transforms: 4 GiB in, 4 GiB out
    const i64 N = 1 << 20;
    cplx *in, *out, *scr;
    CK(hipMalloc(&in, NT * N * 16)); CK(hipMalloc(&out, NT * N * 16)); CK(hipMalloc(&scr, (size_t)3 * 64 * N * 16));
    CK(hipMemset(in, 1, NT * N * 16)); CK(hipMemset(out, 1, NT * N * 16)); CK(hipMemset(scr, 1, (size_t)3 * 64 * N * 16));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));

    if (!*only || !strcmp(only, "xmap")) {
        printf("== XCD mapping granularity, single launch over 256 transforms (TB/s, read+written)\n");
        printf("%6s | %8s %8s %8s %8s\n", "G", "col", "col nt", "row", "row nt");
        int Gs[] = {0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 1024, -1};
        for (int G : Gs) {
            i64 nt = (i64)NT * 128;
            double a = bestms([&] { hipLaunchKernelGGL((k_col<0, 0>), dim3((unsigned)nt), dim3(256), 0, 0, in, out, nt, G); });
            double b = bestms([&] { hipLaunchKernelGGL((k_col<1, 1>), dim3((unsigned)nt), dim3(256), 0, 0, in, out, nt, G); });
            double c = bestms([&] { hipLaunchKernelGGL((k_row<0, 0>), dim3((unsigned)nt), dim3(256), 0, 0, in, out, nt, G); });
            double d = bestms([&] { hipLaunchKernelGGL((k_row<1, 1>), dim3((unsigned)nt), dim3(256), 0, 0, in, out, nt, G); });
            double by = 2.0 * NT * N * 16;
            printf("%6d | %8.2f %8.2f %8.2f %8.2f\n", G, by / a / 1e9, by / b / 1e9, by / c / 1e9, by / d / 1e9);
        }
    }

    if (!*only || !strcmp(only, "chain")) {
        printf("== chain K1 (in -> scratch) ; K2 (scratch -> out) over chunks of C transforms; us per transform\n");
        printf("   policy letters: in-load, scratch-store, scratch-load, out-store (p plain, n nontemporal)\n");
        printf("%5s %6s %4s %8s | %10s %10s %8s\n", "C", "policy", "G", "streams", "us/xform", "TB/s moved", "whole%");
        struct Pol { const char *name; kfn k1, k2; };
        Pol pols[] = {
            {"pppp", k_col<0, 0>, k_row<0, 0>},
            {"nnnn", k_col<1, 1>, k_row<1, 1>},
            {"nppn", k_col<1, 0>, k_row<0, 1>},
            {"pnnp", k_col<0, 1>, k_row<1, 0>},
            {"npnn", k_col<1, 0>, k_row<1, 1>},
        };
        int Cs[] = {8, 16, 24, 32};
        for (int C : Cs) for (auto &p : pols) for (int G : {-1}) for (int streams : {1}) {
            i64 nt = (i64)C * 128;
            if (G > 0 && (nt / 8) % G) continue;
            auto run = [&] {
                int nch = NT / C;
                if (streams == 1) {
                    for (int k = 0; k < nch; ++k) {
                        hipLaunchKernelGGL(p.k1, dim3((unsigned)nt), dim3(256), 0, s1, in + (i64)k * C * N, scr, nt, G);
                        hipLaunchKernelGGL(p.k2, dim3((unsigned)nt), dim3(256), 0, s1, scr, out + (i64)k * C * N, nt, G);
                    }
                } else {
                    // three scratch slots; K1 of chunk k on s1, K2 of chunk k on s2 after an event
                    static hipEvent_t done1[1024], done2[1024]; static bool init = false;
                    if (!init) { for (int i = 0; i < 1024; ++i) { CK(hipEventCreateWithFlags(&done1[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&done2[i], hipEventDisableTiming)); } init = true; }
                    for (int k = 0; k < nch; ++k) {
                        cplx *slot = scr + (i64)(k % 3) * 64 * N;
                        if (k >= 3) CK(hipStreamWaitEvent(s1, done2[k - 3], 0));
                        hipLaunchKernelGGL(p.k1, dim3((unsigned)nt), dim3(256), 0, s1, in + (i64)k * C * N, slot, nt, G);
                        CK(hipEventRecord(done1[k], s1));
                        CK(hipStreamWaitEvent(s2, done1[k], 0));
                        hipLaunchKernelGGL(p.k2, dim3((unsigned)nt), dim3(256), 0, s2, slot, out + (i64)k * C * N, nt, G);
                        CK(hipEventRecord(done2[k], s2));
                    }
                    CK(hipStreamWaitEvent(s1, done2[nch - 1], 0));
                }
            };
            // time on s1 with events
            run(); CK(hipDeviceSynchronize());
            double best = 1e30;
            for (int r = 0; r < 3; ++r) {
                CK(hipEventRecord(e0, s1)); run(); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            CK(hipDeviceSynchronize());
            double us = best * 1e3 / NT;
            printf("%5d %6s %4d %8d | %10.2f %10.2f %8.1f\n", C, p.name, G, streams, us, 4.0 * N * 16 / us / 1e6, 100.0 * 2.0 * N * 16 / us / 1e6 / 8.0);
        }
    }
    return 0;
}
