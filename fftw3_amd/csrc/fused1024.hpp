/*
 * fused1024.hpp -- both passes of an N = 1024 x 1024 transform in ONE launch.
 *
 * Persistent workgroups (two per CU) take tiles in ticket order
 *
 *     unit u:   pass-1 tile i of transform u   |   pass-2 tile i of transform u - LAG     (i = 0..127, alternating)
 *
 * so that a transform's intermediate (16 MiB, scratch slot u % S) is read back
 * by pass 2 a few tens of microseconds after pass 1 wrote it: it is served by
 * the 256 MiB Infinity Cache instead of HBM, and the launch boundary between
 * the passes (and its tail) disappears.  The reference runs the same two steps
 * as ct_apply_dit's child plan and twiddle plan back to back
 * (fftw/fftw_api.c:2078-2090); fusing them is a scheduling change only.
 *
 * Inter-workgroup protocol (MI355X guide section 6, Guideline 16):
 *   producer (pass-1 tile):  all stores -> every wave s_waitcnt vmcnt(0) ->
 *       __syncthreads -> lane 0: agent-scope release fence, s_waitcnt vmcnt(0),
 *       relaxed agent-scope atomic add on done1[transform]
 *   consumer (pass-2 tile):  lane 0 polls done1[transform] == 128 with relaxed
 *       agent-scope loads (+ s_sleep) -> agent-scope acquire fence ->
 *       s_waitcnt vmcnt(0) -> __syncthreads -> plain loads
 * The same pair guards slot reuse: pass 1 of transform u waits for
 * done2[u - S] == 128.  A ticket only waits for lower tickets, and those are
 * held by workgroups that are already running and never block on higher ones,
 * so the grid drains whatever the dispatch order or residency.  Every spin is
 * bounded; on expiry the kernel sets *error and carries on (the host aborts).
 */
#ifndef FA_FUSED1024_HPP
#define FA_FUSED1024_HPP

struct Fused1024Args {
    const double *in;
    double *out;
    double *scratch;          /* S slots of 2^20 complex */
    i64 in_bs, out_bs;        /* batch strides in doubles */
    i64 batch;
    i64 nunits;               /* batch + lag */
    int nslots, lag;
    int *done1;               /* [batch] pass-1 tiles finished */
    int *done2;               /* [batch] pass-2 tiles finished */
    unsigned long long *ticket;
    int *error;
    const cplx *w1024;
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
    int flags;
};

#ifndef FA_FUSED_SPIN_LIMIT
#define FA_FUSED_SPIN_LIMIT (1 << 17)   /* x ~0.3 us: ~40 ms, far beyond any legitimate wait */
#endif

FA_DEV bool fused_wait(int *counter, int want, bool skip = false, int *stat = nullptr) {
    if (skip) return true;
    /* one lane polls; bounded so that a protocol bug cannot hang the GPU */
    for (int spin = 0; spin < FA_FUSED_SPIN_LIMIT; ++spin) {
        if (stat && spin == 1) atomicAdd(stat, 1);          /* waits that did not pass at once */
        if (stat && spin > 0 && (spin & 15) == 0) atomicAdd(stat + 1, 16);
        /* read with a returning RMW: it executes at the memory side, so the value
           is never a stale copy held by this XCD's L2 */
        if (__hip_atomic_fetch_add(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}

/* which counter a ticket depends on (nullptr: none) */
FA_DEV int *fused_dependency(const Fused1024Args &a, unsigned long long tk) {
    const unsigned long long total = (unsigned long long)a.nunits * 256ull;
    if (tk >= total) return nullptr;
    const i64 u = (i64)(tk >> 8);
    const int kind = (int)(tk & 1);
    const i64 b = kind ? u - a.lag : u;
    if (b < 0 || b >= a.batch) return nullptr;
    if (kind) return a.done1 + b;
    return b >= a.nslots ? a.done2 + (b - a.nslots) : nullptr;
}

/* lane 0 fetches the NEXT ticket right after the tile's loads were issued and
   polls that ticket's dependency in the middle of the tile: both round trips
   (~2 us each, memory-side atomics) hide under the tile's own latency */
struct FusedHook {
    const Fused1024Args &a;
    int tid;
    unsigned long long next_ticket;
    int next_poll;
    FA_DEV void after_loads() {
        if (tid == 0) next_ticket = atomicAdd(a.ticket, 1ull);
    }
    /* Polling the next ticket's dependency this early was measured to hurt: the
       poll then runs one tile-time (= two ticket units) sooner, finds the counter
       short far more often and falls into the spin path (LAG 5 / 10 slots:
       15.7 us per transform against 12.3 with the poll at the top of the tile). */
    FA_DEV void mid() {
        if (tid == 0) next_poll = fused_dependency(a, next_ticket) ? -1 : 128;
    }
};

__global__ void __launch_bounds__(256, 2)
fused1024_kernel(const Fused1024Args a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    __shared__ unsigned long long s_ticket;
    __shared__ int s_poll;
    const int tid = threadIdx.x;
    const i64 N1 = 1024;
    const unsigned long long total = (unsigned long long)a.nunits * 256ull;

    if (tid == 0) {
        unsigned long long t0 = atomicAdd(a.ticket, 1ull);
        int *dep = fused_dependency(a, t0);
        s_ticket = t0;
        s_poll = dep ? __hip_atomic_fetch_add(dep, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 128;
    }
    /* the iteration cap is a second safety net: a workgroup can never need more
       tickets than exist */
    for (unsigned long long iter = 0; iter <= total; ++iter) {
        __syncthreads();
        /* wave-uniform by construction: tell the compiler, so that every branch
           below is a scalar branch and no barrier sits under a divergent mask */
        const unsigned long long tkv = s_ticket;
        const unsigned long long tk =
            ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(tkv >> 32)) << 32) |
            (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)tkv);
        const int polled = __builtin_amdgcn_readfirstlane(s_poll);
        __syncthreads();                       /* s_ticket / s_poll are rewritten below */
        if (tk >= total) break;
        const i64 u = (i64)(tk >> 8);
        const int r = (int)(tk & 255);
        const int kind = r & 1, tile = r >> 1;
        const i64 b = kind ? u - a.lag : u;    /* transform this tile belongs to */
        FusedHook hook = { a, tid, 0ull, 128 };
        if (b < 0 || b >= a.batch) {           /* nothing to do for this ticket: just take the next */
            if (tid == 0) {
                unsigned long long t1 = atomicAdd(a.ticket, 1ull);
                int *dep = fused_dependency(a, t1);
                s_ticket = t1;
                s_poll = dep ? __hip_atomic_fetch_add(dep, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 128;
            }
            continue;
        }

        /* dependency: usually already seen satisfied by the prefetched poll */
        if (polled < 128) {
            int *dep = fused_dependency(a, tk);
            if (tid == 0 && dep && !fused_wait(dep, 128, a.flags & (1 << 26),
                                               (a.flags & (1 << 27)) ? a.error + 2 + 2 * kind : nullptr))
                *a.error = 1 + kind;
        }

        P1024Tile t;
        t.w1024 = a.w1024; t.tw_lo = a.tw_lo; t.tw_hi = a.tw_hi; t.tw_shift = a.tw_shift;
        t.Tcur = 8;
        t.lo_sh = 0; t.lo_is = 0; t.lo_os = 0;
        double *slot = a.scratch + (b % a.nslots) * (2 * N1 * N1);
        if (kind == 0) {
            /* pass 1: columns 8*tile .. +7 of the [1024][1024] view, stride 1024 between rows.
               (slot reuse needs no acquire: nothing is read from the slot here) */
            __syncthreads();
            t.src = a.in + b * a.in_bs + (i64)tile * 16;
            t.dst = slot + (i64)tile * 16;
            t.is_l = 2 * N1; t.os_l = 2 * N1;
            t.dis0 = 2; t.dos0 = 2;
            t.dtw0 = 0; t.q0 = 0;
            t.flags = a.flags & FFTW_AMD_F_SWAP_IN;
            p1024_tile<true, true, 0, true, FusedHook>(t, plane, tid, hook);      /* write-through stores */
            /* publish: every storing wave drains its sc1 stores, then one lane signals */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                __hip_atomic_fetch_add(a.done1 + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            /* pass 2: rows 8*tile .. +7 of the intermediate, output transposed into natural order */
            if (tid == 0) {
                if (!(a.flags & (1 << 25))) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            t.src = slot + (i64)tile * 8 * 2 * N1;
            t.dst = a.out + b * a.out_bs + (i64)tile * 16;
            t.is_l = 2; t.os_l = 2 * N1;
            t.dis0 = 2 * N1; t.dos0 = 2;
            t.dtw0 = 1; t.q0 = (i64)tile * 8;
            t.flags = a.flags & FFTW_AMD_F_SWAP_OUT;
            p1024_tile<false, true, 2, false, FusedHook>(t, plane, tid, hook);
            /* the slot may be overwritten once these loads have landed: they have,
               the data went through the butterflies.  No drain of the output stores. */
            if (tid == 0)
                __hip_atomic_fetch_add(a.done2 + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) {
            s_ticket = hook.next_ticket;
            s_poll = hook.next_poll;
        }
    }
}

#endif /* FA_FUSED1024_HPP */
