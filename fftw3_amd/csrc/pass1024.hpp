/*
 * pass1024.hpp -- register-resident 1024-point pass for gfx950 (K1/K2 hot path).
 *
 * One workgroup (256 work-items = 4 wave64) transforms a tile of 8 length-1024
 * sequences.  Every work-item owns 32 elements in registers (128 VGPRs), so the
 * 1024-point DFT is two radix-32 butterflies with ONE exchange through LDS:
 *
 *   l = a + 32 i        item (a,t) holds i = 0..31     -> DFT-32 over i  -> Y_a[d]
 *   Y_a[d] *= conj(w_1024^(a d))                       (t1-codelet twiddles)
 *   exchange through LDS: item (d,t) gets a = 0..31    -> DFT-32 over a  -> X[d + 32 c]
 *   X[l'] *= conj(w_N^(l' q))                          (inter-pass twiddle, optional)
 *
 * This replaces, for n = 1024, what the reference runs as
 * (dft-ct-dit/32 (dftw-direct-32/8 "t2_32") (dft-direct-32-x32 "n1_32"))
 * (fftw/dft_scalar/codelets/n1_32.c, t2_32.c), and the
 * strided column variant replaces dftw-genericbuf (fftw/fftw_api.c:2905-3109).
 *
 * Memory: the 8 sequences of a tile sit side by side either along the
 * contiguous index (column pass: 128-byte segments, stride = row pitch) or are
 * 8 contiguous rows (row pass: 512-byte segments per wave).  IN_T / OUT_T say
 * which index is the fastest across lanes on each side, so loads and stores
 * are 16 B per lane and coalesced in both passes of the N = 2^20 plan.
 *
 * LDS: the exchange moves one real plane at a time (8448 doubles = 66 KiB), so
 * two workgroups fit a CU and overlap each other's HBM latency.  Layouts are
 * conflict-free for ds_write_b64 (16-lane groups) and ds_read_b64 (32-lane
 * groups, 64 banks) in all four IN_T/OUT_T combinations (see lds_index).
 */
#ifndef FA_PASS1024_HPP
#define FA_PASS1024_HPP

/* compile-time (cos, sin)(2 pi m / 32): sin(m) = cos(m - 8) */
template <int M> struct W32 {
    static constexpr double cs[32] = {
        1.0, 0.98078528040323044912618223613423903697393373089333, 0.92387953251128675612818318939678828682241662586364,
        0.83146961230254523707878837761790575673856081198797, 0.70710678118654752440084436210484903928483593768847,
        0.55557023301960222474283081394853287437493719075480, 0.38268343236508977172845998403039886676134456248563,
        0.19509032201612826784828486847702224092769161775195, 0.0, -0.19509032201612826784828486847702224092769161775195,
        -0.38268343236508977172845998403039886676134456248563, -0.55557023301960222474283081394853287437493719075480,
        -0.70710678118654752440084436210484903928483593768847, -0.83146961230254523707878837761790575673856081198797,
        -0.92387953251128675612818318939678828682241662586364, -0.98078528040323044912618223613423903697393373089333,
        -1.0, -0.98078528040323044912618223613423903697393373089333, -0.92387953251128675612818318939678828682241662586364,
        -0.83146961230254523707878837761790575673856081198797, -0.70710678118654752440084436210484903928483593768847,
        -0.55557023301960222474283081394853287437493719075480, -0.38268343236508977172845998403039886676134456248563,
        -0.19509032201612826784828486847702224092769161775195, 0.0, 0.19509032201612826784828486847702224092769161775195,
        0.38268343236508977172845998403039886676134456248563, 0.55557023301960222474283081394853287437493719075480,
        0.70710678118654752440084436210484903928483593768847, 0.83146961230254523707878837761790575673856081198797,
        0.92387953251128675612818318939678828682241662586364, 0.98078528040323044912618223613423903697393373089333
    };
    static constexpr double c = cs[M & 31];
    static constexpr double s = cs[(M + 24) & 31];
};

/* v * conj(w32^M) with the trivial cases folded at compile time */
template <int M> FA_DEV cplx mul_w32c(cplx v) {
    if constexpr ((M & 31) == 0) return v;
    else if constexpr ((M & 31) == 8) return c_mni(v);
    else if constexpr ((M & 31) == 16) return c_make(-v.x, -v.y);
    else if constexpr ((M & 31) == 24) return c_mpi(v);
    else if constexpr ((M & 31) == 4) return c_make((v.x + v.y) * FA_SQRT1_2, (v.y - v.x) * FA_SQRT1_2);
    else if constexpr ((M & 31) == 12) return c_make((v.y - v.x) * FA_SQRT1_2, -(v.x + v.y) * FA_SQRT1_2);
    else if constexpr ((M & 31) == 20) return c_make(-(v.x + v.y) * FA_SQRT1_2, (v.x - v.y) * FA_SQRT1_2);
    else if constexpr ((M & 31) == 28) return c_make((v.x - v.y) * FA_SQRT1_2, (v.x + v.y) * FA_SQRT1_2);
    else return c_mulc(v, c_make(W32<M>::c, W32<M>::s));
}

/* slot of logical output k after bfly32: k = k2 + 8 k1  ->  k1 + 4 k2 */
__host__ __device__ constexpr int slot32(int k) { return (k >> 3) + 4 * (k & 7); }

template <int I> struct Bfly32Step1 {
    static FA_DEV void run(cplx *x) {
        cplx e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = x[I + 4 * j];
        Bfly<8>::run(e);
        x[I] = e[0];
        x[I + 4] = mul_w32c<I * 1>(e[1]);
        x[I + 8] = mul_w32c<I * 2>(e[2]);
        x[I + 12] = mul_w32c<I * 3>(e[3]);
        x[I + 16] = mul_w32c<I * 4>(e[4]);
        x[I + 20] = mul_w32c<I * 5>(e[5]);
        x[I + 24] = mul_w32c<I * 6>(e[6]);
        x[I + 28] = mul_w32c<I * 7>(e[7]);
    }
};

/* forward DFT-32 in place; logical X[k] ends up in x[slot32(k)] */
FA_DEV void bfly32(cplx *x) {
    Bfly32Step1<0>::run(x);
    Bfly32Step1<1>::run(x);
    Bfly32Step1<2>::run(x);
    Bfly32Step1<3>::run(x);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
        cplx c[4] = { x[4 * k2], x[4 * k2 + 1], x[4 * k2 + 2], x[4 * k2 + 3] };
        Bfly<4>::run(c);
        x[4 * k2] = c[0]; x[4 * k2 + 1] = c[1]; x[4 * k2 + 2] = c[2]; x[4 * k2 + 3] = c[3];
    }
}

/* x[slot(d)] *= conj(prefix * w^d) for d in [0,32), w^d built from the five
   binary powers pw[s] = w^(2^s) by a product tree (depth <= 5, 26 complex
   products when prefix == 1).  BIT runs 4..0; D accumulates the index. */
template <int BIT, int D, bool HAVE, bool PERM> struct TwTree {
    static FA_DEV void run(cplx *x, const cplx *pw, cplx acc) {
        TwTree<BIT - 1, D, HAVE, PERM>::run(x, pw, acc);
        cplx nxt = HAVE ? c_mul(acc, pw[BIT]) : pw[BIT];
        TwTree<BIT - 1, D + (1 << BIT), true, PERM>::run(x, pw, nxt);
    }
};
/* PERM: x is in bfly32 output order (index D lives in slot32(D)) */
template <int D, bool HAVE, bool PERM> struct TwTree<-1, D, HAVE, PERM> {
    static FA_DEV void run(cplx *x, const cplx *, cplx acc) {
        constexpr int S = PERM ? slot32(D) : D;
        if (HAVE) x[S] = c_mulc(x[S], acc);
    }
};

/* LDS position (in doubles) of element (d, a, t) of one real plane */
template <bool IN_T, bool OUT_T> FA_DEV int lds_index(int d, int a, int t) {
    if (IN_T && OUT_T) return d * 264 + a * 8 + t;
    if (!IN_T && OUT_T) return d * 264 + t * 33 + a;
    if (IN_T && !OUT_T) return a * 264 + t * 33 + d;
    return t * 1056 + d * 33 + a;
}

#define FA_P1024_LDS_DOUBLES 8448

/* one tile = 8 sequences of 1024; pointers already address the tile origin */
struct P1024Tile {
    const double *src;
    double *dst;
    i64 is_l, os_l;       /* stride of the transform index, in doubles */
    i64 dis0, dos0;       /* stride between the 8 sequences */
    i64 dtw0, q0;         /* twiddle position of sequence t: q0 + t * dtw0 */
    const cplx *w1024;
    const cplx *tw_lo;
    const cplx *tw_hi;
    int tw_shift;
    int Tcur;             /* hi entries present in this tile */
    int flags;
    /* two-level tile dim: sequence t = (t & lo_mask) + (t >> lo_sh) << lo_sh */
    int lo_sh;
    i64 lo_is, lo_os;
    long long *dbg;       /* ABL & 8 only: per-workgroup time stamps */
};

/* offset of sequence t of a tile on the source / destination side */
#define FA_TILE_SOFF(a, t) ((i64)((t) >> (a).lo_sh) * (a).dis0 + (i64)((t) & ((1 << (a).lo_sh) - 1)) * (a).lo_is)
#define FA_TILE_DOFF(a, t) ((i64)((t) >> (a).lo_sh) * (a).dos0 + (i64)((t) & ((1 << (a).lo_sh) - 1)) * (a).lo_os)

/* HAS_TW: 0 none, 1 inter-pass twiddle on the output, 2 on the input.
   `plane` is FA_P1024_LDS_DOUBLES doubles of LDS.  All 256 work-items call. */
/* ABL (tests/micro/p1024_ablate.hip only; 0 in the product): timing ablations that leave
   the memory footprint alone -- 1 no butterflies, 2 no twiddles, 4 no LDS exchange */
template <bool IN_T, bool OUT_T, int HAS_TW, int ABL = 0>
FA_DEV void p1024_tile(const P1024Tile &a, double *plane, const int tid) {
    /* ---- load: item (ai, ti) owns l = ai + 32 i */
    const int ti = IN_T ? (tid & 7) : (tid >> 5);
    const int ai = IN_T ? (tid >> 3) : (tid & 31);
    const int to = OUT_T ? (tid & 7) : (tid >> 5);
    const int dq = OUT_T ? (tid >> 3) : (tid & 31);

    /* ---- every table value this item will need is requested BEFORE the tile's data: the
       vector-memory pipe returns loads in order, so a table load issued after the data has
       been consumed pays a whole round trip through queues that the other workgroups keep
       full (measured with time stamps, tests/micro/p1024_ablate.hip: 4.4 us per tile in
       pass 2, 1 us in pass 1 -- the workgroup life, and with it the pass, is that much longer) */
    cplx pw1024[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) pw1024[s] = (ABL & 2) ? c_make(1.0, 0.0) : a.w1024[ai << s];
    cplx twl[6], twh[6];
    if (HAS_TW != 0 && !(ABL & 2)) {
        const i64 q = a.q0 + (i64)((HAS_TW == 2 ? ti : to) >> a.lo_sh) * a.dtw0;
        const i64 mask = (1LL << a.tw_shift) - 1;
        const i64 m0 = q * (HAS_TW == 2 ? ai : dq);
        twl[5] = a.tw_lo[m0 & mask]; twh[5] = a.tw_hi[m0 >> a.tw_shift];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const i64 m = (q * 32) << s;
            twl[s] = a.tw_lo[m & mask]; twh[s] = a.tw_hi[m >> a.tw_shift];
        }
    }
    cplx x[32];
    {
        const double *p = a.src + (i64)ai * a.is_l + FA_TILE_SOFF(a, ti);
        const i64 step = 32 * a.is_l;
        if ((ti >> a.lo_sh) >= a.Tcur) {
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = c_make(0.0, 0.0);
        } else if (a.flags & FFTW_AMD_F_NT_IN) {
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = ld_cplx<true>(p + i * step);
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = ld_cplx<false>(p + i * step);
        }
        if (a.flags & FFTW_AMD_F_SWAP_IN) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { double s = x[i].x; x[i].x = x[i].y; x[i].y = s; }
        }
    }
    if (ABL & 8) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) a.dbg[1] = wall_clock64();
    }

    /* ---- inter-pass twiddle on the input: conj(w_N^((ai + 32 i) q)) */
    if (HAS_TW == 2 && !(ABL & 2)) {
        cplx base = c_mul(twl[5], twh[5]);
        cplx pw[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) pw[s] = c_mul(twl[s], twh[s]);
        TwTree<4, 0, true, false>::run(x, pw, base);
    }

    /* ---- first radix-32 butterfly over i, then w_1024^(a d) */
    if (!(ABL & 1)) bfly32(x);
    if (!(ABL & 2)) TwTree<4, 0, false, true>::run(x, pw1024, c_make(1.0, 0.0));

    /* ---- exchange through LDS, one real plane at a time */
    cplx y[32];
    if (ABL & 4) {
#pragma unroll
        for (int q = 0; q < 32; ++q) y[q] = x[slot32(q)];
    } else {
#pragma unroll
    for (int d = 0; d < 32; ++d) plane[lds_index<IN_T, OUT_T>(d, ai, ti)] = x[slot32(d)].x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 32; ++q) y[q].x = plane[lds_index<IN_T, OUT_T>(dq, q, to)];
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 32; ++d) plane[lds_index<IN_T, OUT_T>(d, ai, ti)] = x[slot32(d)].y;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 32; ++q) y[q].y = plane[lds_index<IN_T, OUT_T>(dq, q, to)];
    }

    /* ---- second radix-32 butterfly over a: X[dq + 32 c] in y[slot32(c)] */
    if (!(ABL & 1)) bfly32(y);

    /* ---- inter-pass twiddle conj(w_N^((dq + 32 c) q)), q = position of this sequence */
    if (HAS_TW == 1 && !(ABL & 2)) {
        cplx base = c_mul(twl[5], twh[5]);
        cplx pw[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) pw[s] = c_mul(twl[s], twh[s]);
        TwTree<4, 0, true, true>::run(y, pw, base);
    }

    /* ---- store */
    if ((ABL & 8) && tid == 0) a.dbg[2] = wall_clock64();
    if ((to >> a.lo_sh) < a.Tcur) {
        double *p = a.dst + (i64)dq * a.os_l + FA_TILE_DOFF(a, to);
        const i64 step = 32 * a.os_l;
        const bool sw = (a.flags & FFTW_AMD_F_SWAP_OUT) != 0;
        if (a.flags & FFTW_AMD_F_NT_OUT) {
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                cplx v = y[slot32(c)];
                if (sw) { double s = v.x; v.x = v.y; v.y = s; }
                st_cplx<true>(p + c * step, v);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                cplx v = y[slot32(c)];
                if (sw) { double s = v.x; v.x = v.y; v.y = s; }
                st_cplx<false>(p + c * step, v);
            }
        }
    }
    if (ABL & 8) {
        if (tid == 0) a.dbg[3] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) a.dbg[4] = wall_clock64();
        if ((tid & 63) == 0) a.dbg[12 + (tid >> 6)] = wall_clock64();   /* every wave's own end */
    }
}

struct P1024Args {
    const double *src;
    double *dst;
    i64 is_l, os_l;
    i64 dn[FFTW_AMD_MAX_DIMS], dis[FFTW_AMD_MAX_DIMS], dos[FFTW_AMD_MAX_DIMS], dtw[FFTW_AMD_MAX_DIMS];
    const cplx *w1024;
    const cplx *tw_lo;
    const cplx *tw_hi;
    i64 ntiles;
    int tw_shift;
    int ndims, flags;
    int lo_sh;
    i64 lo_is, lo_os;
    long long *dbg;
};

template <bool IN_T, bool OUT_T, int HAS_TW, int ABL = 0>
__global__ void __launch_bounds__(256, 2)
pass1024_kernel(const P1024Args a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    long long t_first = 0;
    if (ABL & 8) t_first = wall_clock64();
    /* workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8): give every XCD one
       contiguous eighth of the tile list, so that the 128-byte segments its CUs touch at
       the same time are neighbours in memory (tests/micro/membw2.hip: +3...15 %) */
    /* the launcher keeps the grid below 2^31 blocks: 32-bit index arithmetic (a 64-bit
       division is ~130 instructions on this ISA, and this prologue is on every workgroup's
       critical path before its first load) */
    unsigned blk = (unsigned)fa_xcd_remap((i64)blockIdx.x, (i64)gridDim.x);
    const unsigned nt = (unsigned)a.ntiles;
    unsigned tile = blk % nt;
    unsigned rest = blk / nt;
    i64 soff = 0, doff = 0, twb = 0;
    for (int d = 1; d < a.ndims; ++d) {
        const unsigned dn = (unsigned)a.dn[d];
        unsigned idx = rest % dn;
        rest /= dn;
        soff += (i64)idx * a.dis[d];
        doff += (i64)idx * a.dos[d];
        twb += (i64)idx * a.dtw[d];
    }
    const i64 t0 = (i64)tile * (8 >> a.lo_sh);
    P1024Tile t;
    t.lo_sh = a.lo_sh; t.lo_is = a.lo_is; t.lo_os = a.lo_os;
    t.src = a.src + soff + t0 * a.dis[0];
    t.dst = a.dst + doff + t0 * a.dos[0];
    t.is_l = a.is_l; t.os_l = a.os_l;
    t.dis0 = a.dis[0]; t.dos0 = a.dos[0];
    t.dtw0 = a.dtw[0]; t.q0 = twb + t0 * a.dtw[0];
    t.w1024 = a.w1024; t.tw_lo = a.tw_lo; t.tw_hi = a.tw_hi; t.tw_shift = a.tw_shift;
    t.Tcur = (int)((a.dn[0] - t0 < (8 >> a.lo_sh)) ? (a.dn[0] - t0) : (8 >> a.lo_sh));
    t.flags = a.flags;
    t.dbg = NULL;
    if (ABL & 8) {
        t.dbg = a.dbg + (i64)blockIdx.x * 16;
        if (threadIdx.x == 0) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            t.dbg[0] = wall_clock64();
            t.dbg[5] = hw; t.dbg[6] = xcc; t.dbg[7] = (long long)blk; t.dbg[8] = t_first;
        }
    }
    p1024_tile<IN_T, OUT_T, HAS_TW, ABL>(t, plane, threadIdx.x);
}

#endif /* FA_PASS1024_HPP */
