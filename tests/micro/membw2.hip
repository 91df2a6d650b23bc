// Micro-benchmark (design evidence, not product): what the MI355X memory system sustains for
// the access shapes of the two 1024-point passes, independent of any FFT arithmetic.
//   * linear float4-style copy in several forms (the guide's 6.29 TB/s figure),
//   * tile copies whose read / write side is made of SEG-element (16 B each) segments at a
//     row pitch, exactly the footprint of pass1024's column loads / transposed stores,
//   * software-pipelined (loads of tile k+1 issued before the stores of tile k) vs phased.
// Build: hipcc -O3 --offload-arch=gfx950 membw2.hip -o membw2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double2 cplx;
typedef long long i64;

template <bool NT> __device__ __forceinline__ cplx ld(const cplx *p) {
    if (NT) { cplx v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(cplx *p, cplx v) {
    if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
    else *p = v;
}

// ---------------------------------------------------------------- linear copies
// every workgroup owns contiguous chunks of THREADS*U elements; persistent grid-stride over chunks
template <int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) cp_chunk(const cplx *__restrict__ s, cplx *__restrict__ d, i64 nchunks) {
    for (i64 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const cplx *sp = s + c * (256 * U) + threadIdx.x;
        cplx *dp = d + c * (256 * U) + threadIdx.x;
        cplx v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NTL>(sp + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) st<NTS>(dp + u * 256, v[u]);
    }
}
template <int U, bool NTL>
__global__ void __launch_bounds__(256) rd_chunk(const cplx *__restrict__ s, cplx *sink, i64 nchunks) {
    cplx acc = {0, 0};
    for (i64 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const cplx *sp = s + c * (256 * U) + threadIdx.x;
        cplx v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NTL>(sp + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
    }
    if (acc.x == 1.2345e300) sink[0] = acc;
}
template <int U, bool NTS>
__global__ void __launch_bounds__(256) wr_chunk(cplx *__restrict__ d, i64 nchunks) {
    for (i64 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        cplx *dp = d + c * (256 * U) + threadIdx.x;
        cplx v = {(double)c, 1.0};
#pragma unroll
        for (int u = 0; u < U; ++u) st<NTS>(dp + u * 256, v);
    }
}

// ---------------------------------------------------------------- tile copies
// A tile has 256*EPT elements.  Side pattern: element e of tile t lives at
//   origin(t) + (e % seg) + (e / seg) * pitch        (seg == tile size: contiguous tile)
// origin(t) = (t % (pitch/seg)) * seg + (t / (pitch/seg)) * (TILE/seg) * pitch
struct Side { int seg_sh; i64 pitch; i64 tiles_per_row_sh; };
template <int EPT> __device__ __forceinline__ i64 origin(const Side &s, i64 t) {
    const i64 TILE = 256 * EPT;
    if ((1 << s.seg_sh) >= TILE) return t * TILE;
    i64 tr = t & ((1LL << s.tiles_per_row_sh) - 1), tb = t >> s.tiles_per_row_sh;
    return (tr << s.seg_sh) + tb * (TILE >> s.seg_sh) * s.pitch;
}
__device__ __forceinline__ i64 eoff(const Side &s, int e) {
    return (e & ((1 << s.seg_sh) - 1)) + (i64)(e >> s.seg_sh) * s.pitch;
}
// XMAP: 0 tile = blockIdx ; 1 consecutive tiles on one XCD (blockIdx%8 = xcd)
template <int XMAP> __device__ __forceinline__ i64 tile_of(i64 blk, i64 ntiles) {
    if (XMAP == 0) return blk;
    i64 per = ntiles >> 3;
    return (blk & 7) * per + (blk >> 3);
}

// phased: all loads, (barrier like the FFT), all stores; one tile per workgroup
template <int EPT, int WGPC, bool NTL, bool NTS, int XMAP, bool BAR>
__global__ void __launch_bounds__(256, WGPC) tile_phased(const cplx *__restrict__ s, cplx *__restrict__ d, Side rs, Side ws, i64 ntiles) {
    i64 t = tile_of<XMAP>(blockIdx.x, ntiles);
    const cplx *sp = s + origin<EPT>(rs, t);
    cplx *dp = d + origin<EPT>(ws, t);
    cplx v[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) v[i] = ld<NTL>(sp + eoff(rs, threadIdx.x + i * 256));
    if (BAR) __syncthreads();
#pragma unroll
    for (int i = 0; i < EPT; ++i) st<NTS>(dp + eoff(ws, threadIdx.x + i * 256), v[i]);
}
// persistent + software pipelined: loads of the next tile are in flight while this tile is stored
template <int EPT, int WGPC, bool NTL, bool NTS, int XMAP>
__global__ void __launch_bounds__(256, WGPC) tile_piped(const cplx *__restrict__ s, cplx *__restrict__ d, Side rs, Side ws, i64 ntiles) {
    i64 k = blockIdx.x;
    if (k >= ntiles) return;
    cplx v[EPT], w[EPT];
    {
        const cplx *sp = s + origin<EPT>(rs, tile_of<XMAP>(k, ntiles));
#pragma unroll
        for (int i = 0; i < EPT; ++i) v[i] = ld<NTL>(sp + eoff(rs, threadIdx.x + i * 256));
    }
    for (;;) {
        i64 kn = k + gridDim.x;
        bool more = kn < ntiles;
        if (more) {
            const cplx *sp = s + origin<EPT>(rs, tile_of<XMAP>(kn, ntiles));
#pragma unroll
            for (int i = 0; i < EPT; ++i) w[i] = ld<NTL>(sp + eoff(rs, threadIdx.x + i * 256));
        }
        cplx *dp = d + origin<EPT>(ws, tile_of<XMAP>(k, ntiles));
#pragma unroll
        for (int i = 0; i < EPT; ++i) st<NTS>(dp + eoff(ws, threadIdx.x + i * 256), v[i]);
        if (!more) break;
#pragma unroll
        for (int i = 0; i < EPT; ++i) v[i] = w[i];
        k = kn;
    }
}

static hipEvent_t e0, e1;
template <class F> static double bestms(F f, int reps = 4) {
    f(); CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}
static Side mkside(int seg, i64 pitch) {
    Side s; s.seg_sh = 0; while ((1 << s.seg_sh) < seg) ++s.seg_sh;
    s.pitch = pitch; s.tiles_per_row_sh = 0;
    if (pitch > 0) { i64 tpr = pitch / seg; while ((1LL << s.tiles_per_row_sh) < tpr) ++s.tiles_per_row_sh; }
    return s;
}

int main(int argc, char **argv) {
    const char *only = argc > 1 ? argv[1] : "";
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t GiB = (size_t)1 << 30;
    size_t bytes = 4 * GiB;                           // per buffer
    cplx *a, *b, *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 1, bytes));
    const i64 n = bytes / 16;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs; buffers 2 x %zu GiB; TB/s = (bytes read + bytes written) / time\n", prop.gcnArchName, cus, bytes / GiB);

    if (!*only || !strcmp(only, "linear")) {
        printf("== linear: contiguous chunk per workgroup, grid = k x CUs (persistent) or one chunk per WG (k=0)\n");
        printf("%-10s %4s %4s %4s | %8s %8s %8s\n", "", "U", "ntl", "nts", "read", "write", "copy");
#define LIN(U, NTL, NTS) do { \
        int ks[] = {0, 4, 8, 16, 32}; \
        for (int k : ks) { \
            i64 nch = n / (256 * U); int grid = k ? k * cus : (int)nch; \
            double r = bestms([&] { hipLaunchKernelGGL((rd_chunk<U, NTL>), dim3(grid), dim3(256), 0, 0, a, sink, nch); }); \
            double w = bestms([&] { hipLaunchKernelGGL((wr_chunk<U, NTS>), dim3(grid), dim3(256), 0, 0, b, nch); }); \
            double c = bestms([&] { hipLaunchKernelGGL((cp_chunk<U, NTL, NTS>), dim3(grid), dim3(256), 0, 0, a, b, nch); }); \
            printf("k=%-8d %4d %4d %4d | %8.2f %8.2f %8.2f\n", k, U, (int)NTL, (int)NTS, bytes / r / 1e9, bytes / w / 1e9, 2.0 * bytes / c / 1e9); \
        } } while (0)
        LIN(1, false, false); LIN(2, false, false); LIN(4, false, false); LIN(8, false, false); LIN(16, false, false);
        LIN(4, true, true); LIN(8, true, true); LIN(4, false, true); LIN(4, true, false);
    }

    if (!*only || !strcmp(only, "tile")) {
        printf("== tile copy, phased (loads | stores), one tile per workgroup; seg in elements of 16 B, pitch 1024\n");
        printf("%-28s %4s %4s %4s %4s %4s | %8s\n", "pattern", "EPT", "wgpc", "nt", "xmap", "bar", "TB/s");
#define TP(EPT, WGPC, NT, XMAP, BAR, RSEG, WSEG, name) do { \
        i64 TILE = 256 * EPT; i64 nt = n / TILE; \
        Side rs = mkside(RSEG ? RSEG : (int)TILE, 1024), ws = mkside(WSEG ? WSEG : (int)TILE, 1024); \
        double c = bestms([&] { hipLaunchKernelGGL((tile_phased<EPT, WGPC, NT, NT, XMAP, BAR>), dim3((unsigned)nt), dim3(256), 0, 0, a, b, rs, ws, nt); }); \
        printf("%-28s %4d %4d %4d %4d %4d | %8.2f\n", name, EPT, WGPC, (int)NT, XMAP, (int)BAR, 2.0 * bytes / c / 1e9); } while (0)
#define TPSET(EPT, WGPC) do { \
        TP(EPT, WGPC, false, 0, true, 0, 0, "contig -> contig"); \
        TP(EPT, WGPC, false, 0, true, 8, 8, "seg8 -> seg8 (pass 1)"); \
        TP(EPT, WGPC, false, 1, true, 8, 8, "seg8 -> seg8 (pass 1)"); \
        TP(EPT, WGPC, true, 0, true, 8, 8, "seg8 -> seg8 (pass 1)"); \
        TP(EPT, WGPC, true, 1, true, 8, 8, "seg8 -> seg8 (pass 1)"); \
        TP(EPT, WGPC, false, 0, true, 0, 8, "contig -> seg8 (pass 2)"); \
        TP(EPT, WGPC, false, 1, true, 0, 8, "contig -> seg8 (pass 2)"); \
        TP(EPT, WGPC, true, 0, true, 0, 8, "contig -> seg8 (pass 2)"); \
        TP(EPT, WGPC, true, 1, true, 0, 8, "contig -> seg8 (pass 2)"); \
        TP(EPT, WGPC, false, 0, true, 16, 16, "seg16 -> seg16"); \
        TP(EPT, WGPC, false, 1, true, 16, 16, "seg16 -> seg16"); \
        TP(EPT, WGPC, false, 0, true, 0, 16, "contig -> seg16"); \
        TP(EPT, WGPC, false, 0, true, 32, 32, "seg32 -> seg32"); \
        TP(EPT, WGPC, false, 0, true, 0, 32, "contig -> seg32"); \
        TP(EPT, WGPC, false, 0, true, 4, 4, "seg4 -> seg4"); \
        TP(EPT, WGPC, false, 0, false, 8, 8, "seg8 -> seg8 no barrier"); \
        TP(EPT, WGPC, false, 1, true, 4, 4, "seg4 -> seg4"); \
        TP(EPT, WGPC, true, 1, true, 4, 4, "seg4 -> seg4"); \
        TP(EPT, WGPC, true, 1, true, 4, 8, "seg4 -> seg8"); \
        TP(EPT, WGPC, true, 1, true, 0, 4, "contig -> seg4"); \
        TP(EPT, WGPC, true, 1, true, 2, 2, "seg2 -> seg2"); \
        TP(EPT, WGPC, true, 1, true, 0, 2, "contig -> seg2"); \
        } while (0)
        TPSET(32, 2); TPSET(16, 4); TPSET(8, 8); TPSET(16, 2); TPSET(32, 1);
    }

    if (!*only || !strcmp(only, "piped")) {
        printf("== tile copy, persistent + pipelined (next tile's loads in flight during stores)\n");
        printf("%-28s %4s %4s %4s %4s %5s | %8s\n", "pattern", "EPT", "wgpc", "nt", "xmap", "grid", "TB/s");
#define PP(EPT, WGPC, NT, XMAP, K, RSEG, WSEG, name) do { \
        i64 TILE = 256 * EPT; i64 nt = n / TILE; \
        Side rs = mkside(RSEG ? RSEG : (int)TILE, 1024), ws = mkside(WSEG ? WSEG : (int)TILE, 1024); \
        int grid = K * cus; \
        double c = bestms([&] { hipLaunchKernelGGL((tile_piped<EPT, WGPC, NT, NT, XMAP>), dim3(grid), dim3(256), 0, 0, a, b, rs, ws, nt); }); \
        printf("%-28s %4d %4d %4d %4d %5d | %8.2f\n", name, EPT, WGPC, (int)NT, XMAP, grid, 2.0 * bytes / c / 1e9); } while (0)
#define PPSET(EPT, WGPC, K) do { \
        PP(EPT, WGPC, false, 0, K, 0, 0, "contig -> contig"); \
        PP(EPT, WGPC, false, 0, K, 8, 8, "seg8 -> seg8 (pass 1)"); \
        PP(EPT, WGPC, true, 0, K, 8, 8, "seg8 -> seg8 (pass 1)"); \
        PP(EPT, WGPC, false, 0, K, 0, 8, "contig -> seg8 (pass 2)"); \
        PP(EPT, WGPC, true, 0, K, 0, 8, "contig -> seg8 (pass 2)"); \
        } while (0)
        PPSET(32, 1, 1); PPSET(16, 2, 2); PPSET(16, 1, 1); PPSET(8, 4, 4); PPSET(8, 2, 2); PPSET(4, 8, 8); PPSET(4, 4, 4);
    }
    return 0;
}
