// Micro-benchmark (design evidence, not product): what the Infinity Cache (256 MiB, memory side)
// gives a two-trip transform on MI355X.  Questions:
//   A. read / write / copy bandwidth as a function of the footprint S (does a resident buffer read
//      or absorb writes faster than HBM, and by how much?)
//   B. a resident scratch of S bytes between a streamed input and a streamed output: the two passes of
//      the 2^20-point plan as linear copies, separate launches, S = 16 MiB ... 1 GiB
//   C. the same with the plan's tile footprints (column tiles / row tiles of membw3.hip) in ONE launch
//      whose block order interleaves pass 1 of transform t with pass 2 of transform t - D (no waits:
//      bandwidth only), so the reuse distance is D transforms whatever the launch size.
// Build: hipcc -O3 --offload-arch=gfx950 mall_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double2 cplx;
typedef long long i64;
typedef double d2 __attribute__((ext_vector_type(2)));

template <int POL> __device__ __forceinline__ cplx ld(const cplx *p) {
    if (POL == 1) { d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p)); cplx r; r.x = v.x; r.y = v.y; return r; }
    return *p;
}
template <int POL> __device__ __forceinline__ void st(cplx *p, cplx v) {
    if (POL == 1) { d2 w = { v.x, v.y }; __builtin_nontemporal_store(w, reinterpret_cast<d2 *>(p)); }
    else *p = v;
}

// ---- A: linear kernels, one 4 KiB piece per workgroup (the fastest copy form on this part, membw2)
template <int LP> __global__ void __launch_bounds__(256) k_read(const cplx *__restrict__ s, cplx *__restrict__ sink, i64 wrap) {
    unsigned i = (blockIdx.x * 256u + threadIdx.x) % (unsigned)wrap;
    cplx v = ld<LP>(s + i);
    if (v.x == 1.2345e300) sink[0] = v;          // never true: keeps the load alive
}
template <int SP> __global__ void __launch_bounds__(256) k_write(cplx *__restrict__ d, i64 wrap) {
    unsigned i = (blockIdx.x * 256u + threadIdx.x) % (unsigned)wrap;
    cplx v; v.x = (double)threadIdx.x; v.y = 1.0;
    st<SP>(d + i, v);
}
template <int LP, int SP> __global__ void __launch_bounds__(256) k_copy(const cplx *__restrict__ s, cplx *__restrict__ d, i64 swrap, i64 dwrap) {
    unsigned g = blockIdx.x * 256u + threadIdx.x;
    st<SP>(d + g % (unsigned)dwrap, ld<LP>(s + g % (unsigned)swrap));
}

// ---- C: the plan's tile footprints.  One launch; block b -> (kind, transform, tile)
// K1 tile: 1024 rows x 8 columns of a [1024][1024] image (128-B segments at a 16 KiB pitch) in -> scratch
// K2 tile: 8 contiguous rows of the scratch image -> 1024 x 8 transposed store into out
template <int P_IN, int P_SS, int P_SL, int P_OUT>
__global__ void __launch_bounds__(256, 2) k_fused(const cplx *__restrict__ in, cplx *__restrict__ scr, cplx *__restrict__ out,
                                                  int NT, int D, int NS) {
    // order: for t = 0 .. NT + D - 1: tiles i = 0..127: [K1(t, i) if t < NT] [K2(t - D, i) if t >= D]
    // blocks per "row" t: 256 in the steady state; head (t < D) 128, tail (t >= NT) 128
    unsigned b = blockIdx.x;
    int t, i, kind;
    const unsigned head = 128u * D;
    if (b < head) { t = b >> 7; i = b & 127; kind = 1; }
    else {
        unsigned r = b - head;
        const unsigned steady = 256u * (NT - D);
        if (r < steady) { t = D + (r >> 8); unsigned w = r & 255; i = w >> 1; kind = (w & 1) ? 2 : 1; }
        else { r -= steady; t = NT + (r >> 7); i = r & 127; kind = 2; }
    }
    cplx v[32];
    const int c = threadIdx.x & 7, rr = threadIdx.x >> 3;
    if (kind == 1) {
        const cplx *sp = in + (i64)t * (1 << 20) + i * 8 + c + (i64)rr * 1024;
        cplx *dp = scr + (i64)(t % NS) * (1 << 20) + i * 8 + c + (i64)rr * 1024;
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = ld<P_IN>(sp + (i64)k * 32 * 1024);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) st<P_SS>(dp + (i64)k * 32 * 1024, v[k]);
    } else {
        const int tt = t - D;
        const cplx *sp = scr + (i64)(tt % NS) * (1 << 20) + i * 8192 + threadIdx.x;
        cplx *dp = out + (i64)tt * (1 << 20) + i * 8 + c + (i64)rr * 1024;
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = ld<P_SL>(sp + k * 256);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) st<P_OUT>(dp + (i64)k * 32 * 1024, v[k]);
    }
}

// ---- D: real launch structures.  One launch = pass-2 tiles of transforms [t2, t2 + n2) and pass-1 tiles of
// transforms [t1, t1 + n1); MIX 0: all pass-2 tiles first (the product's pair launch), 1: alternating
template <int P_IN, int P_SS, int P_SL, int P_OUT>
__global__ void __launch_bounds__(256, 2) k_pair(const cplx *__restrict__ in, cplx *__restrict__ scr, cplx *__restrict__ out,
                                                 int t1, int n1, int t2, int n2, int NS, int mix) {
    unsigned b = blockIdx.x;
    int t, i, kind;
    const unsigned a2 = 128u * n2, a1 = 128u * n1;
    if (mix && n1 == n2) { unsigned w = b >> 1; kind = (b & 1) ? 1 : 2; t = (kind == 1 ? t1 : t2) + (w >> 7); i = w & 127; }
    else if (b < a2) { kind = 2; t = t2 + (b >> 7); i = b & 127; }
    else { b -= a2; kind = 1; t = t1 + (b >> 7); i = b & 127; (void)a1; }
    cplx v[32];
    const int c = threadIdx.x & 7, rr = threadIdx.x >> 3;
    if (kind == 1) {
        const cplx *sp = in + (i64)t * (1 << 20) + i * 8 + c + (i64)rr * 1024;
        cplx *dp = scr + (i64)(t % NS) * (1 << 20) + i * 8 + c + (i64)rr * 1024;
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = ld<P_IN>(sp + (i64)k * 32 * 1024);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) st<P_SS>(dp + (i64)k * 32 * 1024, v[k]);
    } else {
        const cplx *sp = scr + (i64)(t % NS) * (1 << 20) + i * 8192 + threadIdx.x;
        cplx *dp = out + (i64)t * (1 << 20) + i * 8 + c + (i64)rr * 1024;
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = ld<P_SL>(sp + k * 256);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) st<P_OUT>(dp + (i64)k * 32 * 1024, v[k]);
    }
}

// ---- G: what separates the product kernel from the pure copy in the two-lane structure: occupancy (the product's
// 66 KiB LDS plane allows 2 workgroups per CU), the load -> compute -> store phase gap (FMA chain or s_sleep of the
// same length between the last load and the first store)
template <int GAP>   // 0 none, 1 dependent FMA chain, 2 s_sleep
__global__ void __launch_bounds__(256, 2) k_pair_g(const cplx *__restrict__ in, cplx *__restrict__ scr, cplx *__restrict__ out,
                                                   int t1, int n1, int t2, int n2, int NS, int amount) {
    extern __shared__ double lds_dummy[];
    unsigned b = blockIdx.x;
    int t, i, kind;
    const unsigned a2 = 128u * n2;
    if (b < a2) { kind = 2; t = t2 + (b >> 7); i = b & 127; }
    else { b -= a2; kind = 1; t = t1 + (b >> 7); i = b & 127; }
    cplx v[32];
    const int c = threadIdx.x & 7, rr = threadIdx.x >> 3;
    const cplx *sp; cplx *dp;
    if (kind == 1) {
        sp = in + (i64)t * (1 << 20) + i * 8 + c + (i64)rr * 1024;
        dp = scr + (i64)(t % NS) * (1 << 20) + i * 8 + c + (i64)rr * 1024;
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = ld<1>(sp + (i64)k * 32 * 1024);
    } else {
        sp = scr + (i64)(t % NS) * (1 << 20) + i * 8192 + threadIdx.x;
        dp = out + (i64)t * (1 << 20) + i * 8 + c + (i64)rr * 1024;
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = ld<0>(sp + k * 256);
    }
    if (GAP == 1) {
        // every value depends on all loads: a chain through the 32 elements, `amount` rounds of 64 FMAs
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) acc += v[k].x;
        for (int r = 0; r < amount; ++r) {
#pragma unroll
            for (int k = 0; k < 32; ++k) { v[k].x = __builtin_fma(v[k].x, 1.0000001, acc * 1e-300); v[k].y = __builtin_fma(v[k].y, 0.9999999, v[k].x * 1e-300); }
        }
    } else if (GAP == 2) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) acc += v[k].x;
        for (int r = 0; r < amount; ++r) __builtin_amdgcn_s_sleep(64);     // 64 * 64 clocks
        if (acc == 1.2345e300) lds_dummy[threadIdx.x] = acc;
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k].y += acc * 1e-300;
    }
    __syncthreads();
    if (kind == 1) {
#pragma unroll
        for (int k = 0; k < 32; ++k) st<0>(dp + (i64)k * 32 * 1024, v[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) st<1>(dp + (i64)k * 32 * 1024, v[k]);
    }
}

// ---- H: transforms whose intermediate fits an XCD's L2 (n = 2^16: 1 MiB).  ONE launch; the workgroups of XCD x
// (= blockIdx % 8) run pass 1 of their m-th transform followed by pass 2 of their (m - D)-th, the intermediate in a
// per-XCD ring of R slots.  No waits (bandwidth only).  n = 256 x 256: pass-1 tile = 256 rows x 32 columns (512-B
// segments at a 4 KiB pitch), pass-2 tile = 32 contiguous rows in, 256 x 32 transposed out.  LOG2N 15..18.
template <int P_IN, int P_SS, int P_SL, int P_OUT>
__global__ void __launch_bounds__(256, 2) k_xcd(const cplx *__restrict__ in, cplx *__restrict__ scr, cplx *__restrict__ out,
                                                int per_xcd, int D, int R, int log2n, int remap) {
    const unsigned b = blockIdx.x;
    const unsigned x = remap ? (b & 7) : (b / (gridDim.x >> 3));
    const unsigned j = remap ? (b >> 3) : (b % (gridDim.x >> 3));
    const int tiles = 1 << (log2n - 13);                 // tiles of 8192 elements per pass
    const int L2len = 256, L1len = 1 << (log2n - 8);     // n = L1len x 256; pass 1 = columns of the [L1len][256] view
    // per-XCD order: m-th group = [A(m) tiles][B(m - D) tiles]
    const unsigned per = 2u * tiles;
    unsigned m = j / per, w = j % per;
    int kind = w < (unsigned)tiles ? 1 : 2;
    int i = kind == 1 ? w : w - tiles;
    int mm = kind == 1 ? (int)m : (int)m - D;
    if (mm < 0 || mm >= per_xcd) return;
    const i64 n = (i64)1 << log2n;
    const i64 t = (i64)x * per_xcd + mm;                  // transform id: XCD x owns a contiguous range
    const cplx *S = scr + ((i64)x * R + (mm % R)) * n;
    cplx *SW = scr + ((i64)x * R + (mm % R)) * n;
    cplx v[32];
    const int cols = 8192 / L1len;                        // columns per pass-1 tile (32 for 2^16)
    if (kind == 1) {
        // tile i: columns [i*cols, (i+1)*cols) of the [L1len][256] image; thread: c = tid % cols, r0 = tid / cols
        const int c = threadIdx.x % cols, r0 = threadIdx.x / cols, rstep = 256 / cols;
        const cplx *sp = in + t * n + (i64)i * cols + c + (i64)r0 * L2len;
        cplx *dp = SW + (i64)i * cols + c + (i64)r0 * L2len;
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = ld<P_IN>(sp + (i64)k * rstep * L2len);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) st<P_SS>(dp + (i64)k * rstep * L2len, v[k]);
    } else {
        // tile i: 32 contiguous rows of 256 -> transposed store: element (row r, col c) -> out[c * L1len + r]
        const cplx *sp = S + (i64)i * 8192 + threadIdx.x;
        const int r = threadIdx.x & 31, c0 = threadIdx.x >> 5;     // 32 adjacent rows fastest across lanes: 512-B segments
        cplx *dp = out + t * n + (i64)i * 32 + r + (i64)c0 * L1len;
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = ld<P_SL>(sp + k * 256);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) st<P_OUT>(dp + (i64)k * 8 * L1len, v[k]);
    }
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

int main(int argc, char **argv) {
    const char *only = argc > 1 ? argv[1] : "";
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const i64 MiB = 1 << 20;
    const i64 BIG = 8192 * MiB;                   // streamed buffers: 8 GiB each
    cplx *in, *out, *scr;
    CK(hipMalloc(&in, BIG)); CK(hipMalloc(&out, BIG)); CK(hipMalloc(&scr, 2048 * MiB));
    CK(hipMemset(in, 0, BIG)); CK(hipMemset(out, 0, BIG)); CK(hipMemset(scr, 0, 2048 * MiB));
    const i64 sizes[] = { 16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 512, 1024, 2048 };
    const int NSZ = sizeof(sizes) / sizeof(sizes[0]);
    const i64 TOT = 8192 * MiB;                   // bytes moved per timed run (per side)

    if (!*only || strchr(only, 'A')) {
        printf("== A. linear kernels, 4 KiB per workgroup; TB/s of bytes moved; footprint S re-walked until 8 GiB have moved (one launch)\n");
        printf("%8s | %8s %8s | %8s %8s | %10s\n", "S MiB", "read", "read nt", "write", "write nt", "copy S->S'");
        for (int k = 0; k < NSZ; ++k) {
            const i64 S = sizes[k] * MiB, wrap = S / 16;
            const unsigned nb = (unsigned)(TOT / 4096);
            double r0 = bestms([&] { k_read<0><<<nb, 256>>>(scr, out, wrap); });
            double r1 = bestms([&] { k_read<1><<<nb, 256>>>(scr, out, wrap); });
            double w0 = bestms([&] { k_write<0><<<nb, 256>>>(scr, wrap); });
            double w1 = bestms([&] { k_write<1><<<nb, 256>>>(scr, wrap); });
            // copy between two halves of the footprint: S/2 -> S/2
            double c0 = bestms([&] { k_copy<0, 0><<<nb, 256>>>(scr, scr + wrap / 2, wrap / 2, wrap / 2); });
            printf("%8lld | %8.2f %8.2f | %8.2f %8.2f | %10.2f\n", sizes[k], TOT / r0 / 1e9, TOT / r1 / 1e9, TOT / w0 / 1e9,
                   TOT / w1 / 1e9, 2.0 * TOT / c0 / 1e9);
            fflush(stdout);
        }
    }
    if (!*only || strchr(only, 'B')) {
        printf("== B. resident scratch S between streamed in / out, linear copies; 8 GiB in -> scratch, 8 GiB scratch -> out\n");
        printf("   single: one copy alone over 8 GiB (TB/s moved = read + written).  chain: launches alternate in->scr (S bytes), scr->out (S bytes); us per 16 MiB\n");
        printf("%8s | %10s %10s %10s %10s | %12s %12s | %12s %12s\n", "S MiB", "in>scr pp", "in>scr np", "scr>out pp", "scr>out pn",
               "chain pppp", "us/16MiB", "chain nppn", "us/16MiB");
        for (int k = 0; k < NSZ; ++k) {
            const i64 S = sizes[k] * MiB, wrap = S / 16, big = BIG / 16;
            const unsigned nb = (unsigned)(TOT / 4096);
            double a0 = bestms([&] { k_copy<0, 0><<<nb, 256>>>(in, scr, big, wrap); });
            double a1 = bestms([&] { k_copy<1, 0><<<nb, 256>>>(in, scr, big, wrap); });
            double b0 = bestms([&] { k_copy<0, 0><<<nb, 256>>>(scr, out, wrap, big); });
            double b1 = bestms([&] { k_copy<0, 1><<<nb, 256>>>(scr, out, wrap, big); });
            const int nch = (int)(TOT / S);
            const unsigned cb = (unsigned)(S / 4096);
            double c0 = bestms([&] {
                for (int c = 0; c < nch; ++c) {
                    k_copy<0, 0><<<cb, 256>>>(in + (i64)c * wrap, scr, wrap, wrap);
                    k_copy<0, 0><<<cb, 256>>>(scr, out + (i64)c * wrap, wrap, wrap);
                }
            });
            double c1 = bestms([&] {
                for (int c = 0; c < nch; ++c) {
                    k_copy<1, 0><<<cb, 256>>>(in + (i64)c * wrap, scr, wrap, wrap);
                    k_copy<0, 1><<<cb, 256>>>(scr, out + (i64)c * wrap, wrap, wrap);
                }
            });
            printf("%8lld | %10.2f %10.2f %10.2f %10.2f | %12.2f %12.2f | %12.2f %12.2f\n", sizes[k], 2.0 * TOT / a0 / 1e9,
                   2.0 * TOT / a1 / 1e9, 2.0 * TOT / b0 / 1e9, 2.0 * TOT / b1 / 1e9, 4.0 * TOT / c0 / 1e9, c0 * 1e3 / (TOT / (16 * MiB)),
                   4.0 * TOT / c1 / 1e9, c1 * 1e3 / (TOT / (16 * MiB)));
            fflush(stdout);
        }
    }
    if (!*only || strchr(only, 'C')) {
        printf("== C. plan tile footprints, ONE launch, pass 1 of transform t interleaved with pass 2 of transform t - D (no waits); 256 transforms\n");
        printf("   NS = scratch ring in transforms (16 MiB each); us per transform; TB/s = 64 MiB per transform moved\n");
        printf("%4s %4s | %10s %8s | %10s %8s | %10s %8s\n", "D", "NS", "pppp us", "TB/s", "nppn us", "TB/s", "nnnn us", "TB/s");
        const int NT = 256;
        const int Ds[] = { 1, 2, 3, 4, 6, 8, 12, 16, 32, 64 };
        for (unsigned k = 0; k < sizeof(Ds) / sizeof(Ds[0]); ++k) {
            const int D = Ds[k];
            for (int extra = 0; extra < 2; ++extra) {
                const int NS = extra ? 128 : D + 4;
                if (extra && D != 4 && D != 16) continue;
                const unsigned nb = 256u * NT;
                double t0 = bestms([&] { k_fused<0, 0, 0, 0><<<nb, 256>>>(in, scr, out, NT, D, NS); });
                double t1 = bestms([&] { k_fused<1, 0, 0, 1><<<nb, 256>>>(in, scr, out, NT, D, NS); });
                double t2 = bestms([&] { k_fused<1, 1, 1, 1><<<nb, 256>>>(in, scr, out, NT, D, NS); });
                const double by = 64.0 * MiB * NT;
                printf("%4d %4d | %10.2f %8.2f | %10.2f %8.2f | %10.2f %8.2f\n", D, NS, t0 * 1e3 / NT, by / t0 / 1e9, t1 * 1e3 / NT,
                       by / t1 / 1e9, t2 * 1e3 / NT, by / t2 / 1e9);
                fflush(stdout);
            }
        }
    }
    if (!*only || strchr(only, 'D')) {
        printf("== D. launch structures (no arithmetic), 256 transforms, policy nppn (pppp in brackets); us per transform\n");
        printf("   one stream: launch k = pass 2 of chunk k-1 + pass 1 of chunk k.  two streams: launch k on stream k%%2 = pass 2 of chunk k-2 + pass 1 of chunk k\n");
        printf("%4s | %22s %22s | %22s %22s\n", "C", "1 stream, p2 then p1", "1 stream, alternating", "2 streams, p2 then p1", "2 streams, alternating");
        const int NT = 256;
        hipStream_t st[2];
        CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
        hipEvent_t eb, ee[2];
        CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ee[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ee[1], hipEventDisableTiming));
        const int Cs[] = { 1, 2, 4, 8, 16, 32 };
        for (unsigned k = 0; k < sizeof(Cs) / sizeof(Cs[0]); ++k) {
            const int C = Cs[k], nch = NT / C;
            double r[8];
            for (int pol = 0; pol < 2; ++pol)
                for (int two = 0; two < 2; ++two)
                    for (int mix = 0; mix < 2; ++mix) {
                        const int lag = two ? 2 : 1, NS = 2 * lag * C;   // a slot is rewritten only after the launch that read it has completed
                        r[pol * 4 + two * 2 + mix] = bestms([&] {
                            if (two) { CK(hipEventRecord(eb, 0)); CK(hipStreamWaitEvent(st[0], eb, 0)); CK(hipStreamWaitEvent(st[1], eb, 0)); }
                            for (int l = 0; l < nch + lag; ++l) {
                                const int n1 = l < nch ? C : 0, n2 = l >= lag ? C : 0;
                                hipStream_t s_ = two ? st[l & 1] : (hipStream_t)0;
                                const unsigned nb = 128u * (n1 + n2);
                                if (pol) k_pair<1, 0, 0, 1><<<nb, 256, 0, s_>>>(in, scr, out, l * C, n1, (l - lag) * C, n2, NS, mix);
                                else k_pair<0, 0, 0, 0><<<nb, 256, 0, s_>>>(in, scr, out, l * C, n1, (l - lag) * C, n2, NS, mix);
                            }
                            if (two) { CK(hipEventRecord(ee[0], st[0])); CK(hipEventRecord(ee[1], st[1])); CK(hipStreamWaitEvent(0, ee[0], 0)); CK(hipStreamWaitEvent(0, ee[1], 0)); }
                        });
                    }
            printf("%4d | %10.2f (%8.2f) %10.2f (%8.2f) | %10.2f (%8.2f) %10.2f (%8.2f)\n", C, r[4] * 1e3 / NT, r[0] * 1e3 / NT, r[5] * 1e3 / NT,
                   r[1] * 1e3 / NT, r[6] * 1e3 / NT, r[2] * 1e3 / NT, r[7] * 1e3 / NT, r[3] * 1e3 / NT);
            fflush(stdout);
        }
    }
    if (!*only || strchr(only, 'E')) {
        printf("== E. S streams, launch k on stream k%%S = pass 2 of chunk k-S + pass 1 of chunk k (ring of 2*S*C transforms), policy nppn, 256 transforms; us per transform (alternating tiles / p2 then p1)\n");
        const int NT = 240;     // divisible by 1..6, 8
        hipStream_t st[4];
        hipEvent_t eb, ee[4];
        CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
        for (int i = 0; i < 4; ++i) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ee[i], hipEventDisableTiming)); }
        const int Cs[] = { 1, 2, 3, 4, 6, 8 };
        printf("%4s |", "C");
        for (int S = 1; S <= 4; ++S) printf("        S=%d          |", S);
        printf("\n");
        for (unsigned k = 0; k < sizeof(Cs) / sizeof(Cs[0]); ++k) {
            const int C = Cs[k], nch = NT / C;
            printf("%4d |", C);
            for (int S = 1; S <= 4; ++S) {
                double r[2];
                for (int mix = 0; mix < 2; ++mix) {
                    const int lag = S, NS = 2 * lag * C;
                    r[mix] = bestms([&] {
                        CK(hipEventRecord(eb, 0));
                        for (int i = 0; i < S; ++i) CK(hipStreamWaitEvent(st[i], eb, 0));
                        for (int l = 0; l < nch + lag; ++l) {
                            const int n1 = l < nch ? C : 0, n2 = l >= lag ? C : 0;
                            const unsigned nb = 128u * (n1 + n2);
                            k_pair<1, 0, 0, 1><<<nb, 256, 0, st[l % S]>>>(in, scr, out, l * C, n1, (l - lag) * C, n2, NS, mix);
                        }
                        for (int i = 0; i < S; ++i) { CK(hipEventRecord(ee[i], st[i])); CK(hipStreamWaitEvent(0, ee[i], 0)); }
                    });
                }
                printf(" %8.2f / %8.2f |", r[1] * 1e3 / NT, r[0] * 1e3 / NT);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    if (!*only || strchr(only, 'F')) {
        printf("== F. S independent streams: chunk c runs pass 1 then pass 2 on stream c%%S with its own scratch slot (ring of S*C transforms), policy nppn (pppp), 240 transforms; us per transform\n");
        const int NT = 240;
        hipStream_t st[6];
        hipEvent_t eb, ee[6];
        CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
        for (int i = 0; i < 6; ++i) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ee[i], hipEventDisableTiming)); }
        const int Cs[] = { 1, 2, 3, 4, 6, 8, 16 };
        printf("%4s |", "C");
        for (int S = 1; S <= 6; ++S) printf("        S=%d          |", S);
        printf("\n");
        for (unsigned k = 0; k < sizeof(Cs) / sizeof(Cs[0]); ++k) {
            const int C = Cs[k], nch = NT / C;
            printf("%4d |", C);
            for (int S = 1; S <= 6; ++S) {
                double r[2];
                for (int pol = 0; pol < 2; ++pol) {
                    const int NS = S * C;
                    r[pol] = bestms([&] {
                        CK(hipEventRecord(eb, 0));
                        for (int i = 0; i < S; ++i) CK(hipStreamWaitEvent(st[i], eb, 0));
                        for (int l = 0; l < nch; ++l) {
                            const unsigned nb = 128u * C;
                            if (pol) {
                                k_pair<1, 0, 0, 1><<<nb, 256, 0, st[l % S]>>>(in, scr, out, l * C, C, 0, 0, NS, 0);
                                k_pair<1, 0, 0, 1><<<nb, 256, 0, st[l % S]>>>(in, scr, out, 0, 0, l * C, C, NS, 0);
                            } else {
                                k_pair<0, 0, 0, 0><<<nb, 256, 0, st[l % S]>>>(in, scr, out, l * C, C, 0, 0, NS, 0);
                                k_pair<0, 0, 0, 0><<<nb, 256, 0, st[l % S]>>>(in, scr, out, 0, 0, l * C, C, NS, 0);
                            }
                        }
                        for (int i = 0; i < S; ++i) { CK(hipEventRecord(ee[i], st[i])); CK(hipStreamWaitEvent(0, ee[i], 0)); }
                    });
                }
                printf(" %8.2f (%8.2f) |", r[1] * 1e3 / NT, r[0] * 1e3 / NT);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    if (!*only || strchr(only, 'G')) {
        printf("== G. two lanes, chunks of 8 transforms (structure F, S=2, C=8), nppn: occupancy and the phase gap; us per transform\n");
        const int NT = 240, C = 8, S = 2, NS = S * C, nch = NT / C;
        hipStream_t st[2];
        hipEvent_t eb, ee[2];
        CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ee[i], hipEventDisableTiming)); }
        CK(hipFuncSetAttribute((const void *)k_pair_g<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10));
        CK(hipFuncSetAttribute((const void *)k_pair_g<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10));
        CK(hipFuncSetAttribute((const void *)k_pair_g<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10));
        struct { int gap, amount; const char *name; } V[] = {
            { 0, 0, "no gap" }, { 1, 5, "FMA chain 320/item (~0.55 us)" }, { 1, 10, "FMA chain 640/item" }, { 1, 20, "FMA chain 1280/item (~2.1 us, the FFT's arithmetic)" },
            { 1, 30, "FMA chain 1920/item" }, { 2, 1, "s_sleep 4096 clk (~1.7 us)" }, { 2, 2, "s_sleep 8192 clk" } };
        const size_t L[] = { 0, 40 << 10, 53 << 10, 66 << 10 };
        printf("%-52s | %10s %10s %10s %10s\n", "gap \\ dynamic LDS per workgroup (WG/CU)", "0 (regs)", "40K (4)", "53K (3)", "66K (2)");
        for (unsigned v = 0; v < sizeof(V) / sizeof(V[0]); ++v) {
            printf("%-52s |", V[v].name);
            for (int l = 0; l < 4; ++l) {
                double r = bestms([&] {
                    CK(hipEventRecord(eb, 0));
                    for (int i = 0; i < S; ++i) CK(hipStreamWaitEvent(st[i], eb, 0));
                    for (int c = 0; c < nch; ++c) {
                        const unsigned nb = 128u * C;
                        for (int pass = 0; pass < 2; ++pass) {
                            const int n1 = pass ? 0 : C, n2 = pass ? C : 0;
                            if (V[v].gap == 0) k_pair_g<0><<<nb, 256, L[l], st[c % S]>>>(in, scr, out, c * C, n1, c * C, n2, NS, 0);
                            else if (V[v].gap == 1) k_pair_g<1><<<nb, 256, L[l], st[c % S]>>>(in, scr, out, c * C, n1, c * C, n2, NS, V[v].amount);
                            else k_pair_g<2><<<nb, 256, L[l], st[c % S]>>>(in, scr, out, c * C, n1, c * C, n2, NS, V[v].amount);
                        }
                    }
                    for (int i = 0; i < S; ++i) { CK(hipEventRecord(ee[i], st[i])); CK(hipStreamWaitEvent(0, ee[i], 0)); }
                });
                printf(" %10.2f", r * 1e3 / NT);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    if (!*only || strchr(only, 'H')) {
        printf("== H. intermediate in the XCD's L2: one launch, XCD x runs pass 1 of its m-th transform then pass 2 of its (m-D)-th; 4 GiB in, 4 GiB out; no arithmetic, no waits\n");
        printf("   us per transform and TB/s on the ALGORITHMIC bytes (32 n per transform); 'two launches' = pass 1 over everything, then pass 2 (scratch = 4 GiB)\n");
        printf("%6s %3s %3s %6s | %10s %8s | %10s %8s | %10s %8s\n", "log2n", "D", "R", "remap", "pppp us", "TB/s", "nppn us", "TB/s", "nnnn us", "TB/s");
        for (int log2n = 15; log2n <= 18; ++log2n) {
            const i64 n = (i64)1 << log2n;
            const int NT = (int)((4096 * MiB) / (16 * n));      // transforms in 4 GiB
            const int per_xcd = NT / 8, tiles = 1 << (log2n - 13);
            const double alg = 32.0 * n * NT;
            // two launches: D = per_xcd (all pass 1 first), ring = everything
            struct { int D, R, remap; } V[] = { { 1, 2, 1 }, { 2, 3, 1 }, { 2, 4, 1 }, { 4, 6, 1 }, { 1, 2, 0 }, { per_xcd, per_xcd, 1 } };
            for (unsigned k = 0; k < sizeof(V) / sizeof(V[0]); ++k) {
                const int D = V[k].D, R = V[k].R;
                if ((i64)8 * R * n * 16 > 2048 * MiB && D != per_xcd) continue;
                cplx *scrp = D == per_xcd ? out : scr;          // the two-launch form needs a 4 GiB scratch: borrow the upper half of `out`
                cplx *outp = out;
                if (D == per_xcd) scrp = out + (i64)NT * n;     // out is 8 GiB: second half
                const unsigned nb = 8u * (unsigned)(per_xcd + D) * 2u * tiles;
                double t0 = bestms([&] { k_xcd<0, 0, 0, 0><<<nb, 256>>>(in, scrp, outp, per_xcd, D, R, log2n, V[k].remap); });
                double t1 = bestms([&] { k_xcd<1, 0, 0, 1><<<nb, 256>>>(in, scrp, outp, per_xcd, D, R, log2n, V[k].remap); });
                double t2 = bestms([&] { k_xcd<1, 1, 1, 1><<<nb, 256>>>(in, scrp, outp, per_xcd, D, R, log2n, V[k].remap); });
                printf("%6d %3d %3d %6d | %10.3f %8.2f | %10.3f %8.2f | %10.3f %8.2f%s\n", log2n, D == per_xcd ? -1 : D, D == per_xcd ? -1 : R, V[k].remap,
                       t0 * 1e3 / NT, alg / t0 / 1e9, t1 * 1e3 / NT, alg / t1 / 1e9, t2 * 1e3 / NT, alg / t2 / 1e9, D == per_xcd ? "   (all of pass 1, then all of pass 2)" : "");
                fflush(stdout);
            }
        }
    }
    return 0;
}
