// Micro-benchmark (design evidence, not product): Infinity-Cache vs HBM bandwidth on MI355X,
// alone and concurrently.  Build: hipcc -O3 --offload-arch=gfx950 membw.hip -o membw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// each block streams `per_block` bytes starting at (blockIdx * per_block) % span, `reps` sweeps
__global__ void __launch_bounds__(256) rd(const double2* __restrict__ p, size_t span_elems, size_t total_elems, double2* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    double2 acc = {0, 0};
    const size_t mask = span_elems - 1;     // spans are powers of two
    for (; i + 7 * stride < total_elems; i += 8 * stride) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(i + u * stride) & mask];
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
    }
    if (acc.x == 1.2345e300) sink[0] = acc;
}
__global__ void __launch_bounds__(256) wr(double2* __restrict__ p, size_t span_elems, size_t total_elems) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t mask = span_elems - 1;
    for (; i < total_elems; i += stride) {
        double2 v = {(double)i, 1.0};
        p[i & mask] = v;
    }
}
__global__ void __launch_bounds__(256) cp(const double2* __restrict__ s, size_t sspan, double2* __restrict__ d, size_t dspan, size_t total_elems) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t sm = sspan - 1, dm = dspan - 1;
    for (; i + 7 * stride < total_elems; i += 8 * stride) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s[(i + u * stride) & sm];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[(i + u * stride) & dm] = v[u];
    }
}
static float timeit(hipStream_t st, void (*f)(hipStream_t, void*), void* a) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(st, a); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); f(st, a); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}
int main() {
    size_t big = (size_t)8 << 30;           // 8 GiB buffer
    double2 *a, *b, *sink;
    CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, big)); CK(hipMemset(b, 1, big));
    size_t total = ((size_t)8 << 30) / 16;  // elements moved per launch (8 GiB)
    int grid = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    size_t spans_mib[] = {16, 32, 64, 128, 256, 512, 8192};
    printf("read-only, working set cycling (8 GiB moved):\n");
    for (size_t s : spans_mib) {
        size_t span = (s << 20) / 16;
        hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, a, span, total, sink); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, a, span, total, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  span %5zu MiB: %.2f TB/s\n", s, total * 16.0 / ms / 1e9);
    }
    printf("write-only:\n");
    for (size_t s : spans_mib) {
        size_t span = (s << 20) / 16;
        hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, b, span, total); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, b, span, total); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  span %5zu MiB: %.2f TB/s\n", s, total * 16.0 / ms / 1e9);
    }
    printf("copy (read span / write span), bytes = read+write:\n");
    size_t combos[][2] = {{8192, 8192}, {64, 8192}, {8192, 64}, {64, 64}, {128, 128}, {16, 16}};
    for (auto& c : combos) {
        size_t ss = (c[0] << 20) / 16, ds = (c[1] << 20) / 16;
        hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, a, ss, b, ds, total / 2); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, a, ss, b, ds, total / 2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  rd %5zu MiB -> wr %5zu MiB: %.2f TB/s total\n", c[0], c[1], total * 16.0 / ms / 1e9);
    }
    // concurrent: stream 1 reads a 64 MiB resident span, stream 2 streams 8 GiB from HBM
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    printf("concurrent read: resident 64 MiB span on stream 1 + HBM stream on stream 2 (half the grid each):\n");
    for (int rep = 0; rep < 2; ++rep) {
        size_t span = ((size_t)64 << 20) / 16;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0)); CK(hipStreamWaitEvent(s1, e0, 0)); CK(hipStreamWaitEvent(s2, e0, 0));
        hipLaunchKernelGGL(rd, dim3(grid / 2), dim3(256), 0, s1, a, span, total, sink);
        hipLaunchKernelGGL(rd, dim3(grid / 2), dim3(256), 0, s2, b, big / 16, total, sink);
        hipEvent_t f1, f2; CK(hipEventCreate(&f1)); CK(hipEventCreate(&f2));
        CK(hipEventRecord(f1, s1)); CK(hipEventRecord(f2, s2));
        CK(hipEventSynchronize(f1)); CK(hipEventSynchronize(f2));
        float m1, m2; CK(hipEventElapsedTime(&m1, e0, f1)); CK(hipEventElapsedTime(&m2, e0, f2));
        printf("  resident done %.3f ms (%.2f TB/s), hbm done %.3f ms (%.2f TB/s); aggregate over max: %.2f TB/s\n",
               m1, total * 16.0 / m1 / 1e9, m2, total * 16.0 / m2 / 1e9, 2 * total * 16.0 / (m1 > m2 ? m1 : m2) / 1e9);
    }
    return 0;
}
