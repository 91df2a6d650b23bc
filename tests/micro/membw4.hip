// Micro-benchmark (design evidence, not product): the two-pass chain of membw3 with a stand-in
// for the FFT arithmetic (a dependent FP64 FMA chain of FMAS instructions per work-item between
// the loads and the stores), to see how much of it the memory phases of the other workgroups
// hide when a launch is only a few waves of workgroups long ("convoy": every workgroup of a
// short launch loads, computes and stores at the same time), and whether a staggered start of
// the first wave of workgroups repairs it.   Build: hipcc -O3 --offload-arch=gfx950 membw4.hip -o membw4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double2 cplx;
typedef long long i64;
template <int POL> __device__ __forceinline__ cplx ld(const cplx *p) {
    if (POL == 1) { cplx v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
    return *p;
}
template <int POL> __device__ __forceinline__ void st(cplx *p, cplx v) {
    if (POL == 1) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
    else *p = v;
}
__device__ __forceinline__ i64 tile_of(i64 blk, i64 ntiles) { return (blk & 7) * (ntiles >> 3) + (blk >> 3); }
struct Knobs { int fmas; int stagger_ns; int first_wave; };
__device__ __forceinline__ void stagger(const Knobs &k) {
    // first_wave > 0: delay one of the two workgroups that (presumably) share a CU by stagger_ns.
    // mode (k.first_wave >> 16): 0 = slot parity within the XCD (j & 1), 1 = second round of 32 CUs ((j / 32) & 1)
    const int fw = k.first_wave & 0xffff, mode = k.first_wave >> 16;
    if (k.stagger_ns > 0 && (int)blockIdx.x < fw) {
        int j = blockIdx.x >> 3;
        int ph = mode == 0 ? (j & 1) : ((j >> 5) & 1);
        if (ph) {
            long long t0 = wall_clock64();                     // 100 MHz
            long long wait = (long long)k.stagger_ns / 10;
            while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
        }
    }
}
__device__ __forceinline__ void fake_compute(cplx *v, int fmas) {
    // 64 independent chains (the 32 complex values), fmas/64 rounds: VALU-bound like the butterflies
    for (int r = 0; r < fmas / 64; ++r) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { v[i].x = fma(v[i].x, 1.0000001, 1e-9); v[i].y = fma(v[i].y, 0.9999999, -1e-9); }
    }
}
template <int LP, int SP, int W>
__global__ void __launch_bounds__(256, W) k_col(const cplx *__restrict__ s, cplx *__restrict__ d, i64 ntiles, Knobs kn) {
    extern __shared__ double ldsbuf[];
    if (kn.fmas < 0) ldsbuf[threadIdx.x] = 0;
    stagger(kn);
    i64 t = tile_of(blockIdx.x, ntiles);
    i64 org = (t >> 7) * (1 << 20) + (t & 127) * 8;
    const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
    const cplx *sp = s + org + c + (i64)r * 1024;
    cplx *dp = d + org + c + (i64)r * 1024;
    cplx v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = ld<LP>(sp + (i64)i * 32 * 1024);
    fake_compute(v, kn.fmas);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; ++i) st<SP>(dp + (i64)i * 32 * 1024, v[i]);
}
template <int LP, int SP, int W>
__global__ void __launch_bounds__(256, W) k_row(const cplx *__restrict__ s, cplx *__restrict__ d, i64 ntiles, Knobs kn) {
    extern __shared__ double ldsbuf[];
    if (kn.fmas < 0) ldsbuf[threadIdx.x] = 0;
    stagger(kn);
    i64 t = tile_of(blockIdx.x, ntiles);
    const cplx *sp = s + t * 8192 + threadIdx.x;
    i64 org = (t >> 7) * (1 << 20) + (t & 127) * 8;
    const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
    cplx *dp = d + org + c + (i64)r * 1024;
    cplx v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = ld<LP>(sp + i * 256);
    fake_compute(v, kn.fmas);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; ++i) st<SP>(dp + (i64)i * 32 * 1024, v[i]);
}
typedef void (*kfn)(const cplx *, cplx *, i64, Knobs);
int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NT = 256; const i64 N = 1 << 20;
    cplx *in, *out, *scr;
    CK(hipMalloc(&in, NT * N * 16)); CK(hipMalloc(&out, NT * N * 16)); CK(hipMalloc(&scr, (size_t)64 * N * 16));
    CK(hipMemset(in, 0, NT * N * 16)); CK(hipMemset(out, 0, NT * N * 16)); CK(hipMemset(scr, 0, (size_t)64 * N * 16));
    hipStream_t s1; CK(hipStreamCreate(&s1));
    printf("chain K1 ; K2 over chunks of C transforms, XCD-contiguous tiles; fmas = dependent-chain FP64 FMAs per work-item\n");
    printf("(the real pass has ~1300-1700 FP64 VALU instructions per work-item); stagger = phase step of the first 512 workgroups\n");
    printf("%5s %6s %6s %8s | %10s %8s\n", "C", "policy", "fmas", "stagger", "us/xform", "whole%");
    struct Pol { const char *name; kfn k1, k2; };
    Pol pols[] = { {"nppn", k_col<1, 0, 1>, k_row<0, 1, 1>} };
    CK(hipFuncSetAttribute((const void *)k_col<1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_row<0, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (auto &p : pols) for (int ldskb : {80}) for (int C : {16}) for (int fmas : {1280}) for (int mode : {0, 1}) for (int stg : {0, 2000, 4000, 6000, 8000, 12000}) {
        const size_t lds = (size_t)ldskb * 1024;
        i64 nt = (i64)C * 128;
        Knobs kn; kn.fmas = fmas; kn.stagger_ns = stg; kn.first_wave = 512 | (mode << 16);
        auto run = [&] {
            for (int k = 0; k < NT / C; ++k) {
                hipLaunchKernelGGL(p.k1, dim3((unsigned)nt), dim3(256), lds, s1, in + (i64)k * C * N, scr, nt, kn);
                hipLaunchKernelGGL(p.k2, dim3((unsigned)nt), dim3(256), lds, s1, scr, out + (i64)k * C * N, nt, kn);
            }
        };
        run(); CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, s1)); run(); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double us = best * 1e3 / NT;
        printf("%5d %8s %6d mode %d stagger %5d | %10.2f %8.1f\n", C, p.name, fmas, mode, stg, us, 100.0 * 2.0 * N * 16 / us / 1e6 / 8.0);
        fflush(stdout);
    }
    return 0;
}
