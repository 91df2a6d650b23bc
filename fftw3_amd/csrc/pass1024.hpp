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
};

/* offset of sequence t of a tile on the source / destination side */
#define FA_TILE_SOFF(a, t) ((i64)((t) >> (a).lo_sh) * (a).dis0 + (i64)((t) & ((1 << (a).lo_sh) - 1)) * (a).lo_is)
#define FA_TILE_DOFF(a, t) ((i64)((t) >> (a).lo_sh) * (a).dos0 + (i64)((t) & ((1 << (a).lo_sh) - 1)) * (a).lo_os)

/* HAS_TW: 0 none, 1 inter-pass twiddle on the output, 2 on the input.
   `plane` is FA_P1024_LDS_DOUBLES doubles of LDS.  All 256 work-items call. */
typedef unsigned int fa_u32x4 __attribute__((ext_vector_type(4)));

/* ST_SC1: the tile is handed to another workgroup inside the launch, so its
   stores are write-through (sc1) buffer stores (guide section 6 G16, form R1) */
/* hooks let a persistent caller slip independent scalar work (next ticket,
   dependency poll) under the tile's memory latency */
struct P1024NoHook {
    FA_DEV void after_loads() {}
    FA_DEV void mid() {}
};

template <bool IN_T, bool OUT_T, int HAS_TW, bool ST_SC1, class Hook>
FA_DEV void p1024_tile(const P1024Tile &a, double *plane, const int tid, Hook &hook) {
    /* ---- load: item (ai, ti) owns l = ai + 32 i */
    const int ti = IN_T ? (tid & 7) : (tid >> 5);
    const int ai = IN_T ? (tid >> 3) : (tid & 31);
    cplx x[32];
    {
        const double *p = a.src + (i64)ai * a.is_l + FA_TILE_SOFF(a, ti);
        const i64 step = 32 * a.is_l;
        if ((ti >> a.lo_sh) < a.Tcur) {
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = *reinterpret_cast<const cplx *>(p + i * step);
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = c_make(0.0, 0.0);
        }
        hook.after_loads();
        if (a.flags & FFTW_AMD_F_SWAP_IN) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { double s = x[i].x; x[i].x = x[i].y; x[i].y = s; }
        }
    }

    /* ---- inter-pass twiddle on the input: conj(w_N^((ai + 32 i) q)) */
    if (HAS_TW == 2) {
        const i64 q = a.q0 + (i64)(ti >> a.lo_sh) * a.dtw0;
        cplx base = tw2(a.tw_lo, a.tw_hi, a.tw_shift, q * ai);
        cplx pw[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) pw[s] = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (q * 32) << s);
        TwTree<4, 0, true, false>::run(x, pw, base);
    }

    /* ---- first radix-32 butterfly over i, then w_1024^(a d) */
    bfly32(x);
    {
        cplx pw[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) pw[s] = a.w1024[ai << s];
        TwTree<4, 0, false, true>::run(x, pw, c_make(1.0, 0.0));
    }

    /* ---- exchange through LDS, one real plane at a time */
    const int to = OUT_T ? (tid & 7) : (tid >> 5);
    const int dq = OUT_T ? (tid >> 3) : (tid & 31);
    cplx y[32];
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

    hook.mid();

    /* ---- second radix-32 butterfly over a: X[dq + 32 c] in y[slot32(c)] */
    bfly32(y);

    /* ---- inter-pass twiddle conj(w_N^((dq + 32 c) q)), q = position of this sequence */
    if (HAS_TW == 1) {
        const i64 q = a.q0 + (i64)(to >> a.lo_sh) * a.dtw0;
        cplx base = tw2(a.tw_lo, a.tw_hi, a.tw_shift, q * dq);
        cplx pw[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) pw[s] = tw2(a.tw_lo, a.tw_hi, a.tw_shift, (q * 32) << s);
        TwTree<4, 0, true, true>::run(y, pw, base);
    }

    /* ---- store */
    if (ST_SC1) {
        /* wave-uniform descriptor over the tile's destination; per-lane byte offsets */
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, 0x7fffffff, 0x00020000);
        const int off0 = (int)(((i64)dq * a.os_l + FA_TILE_DOFF(a, to)) * 8);
        const int step = (int)(32 * a.os_l * 8);
        if ((to >> a.lo_sh) < a.Tcur) {
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                cplx v = y[slot32(c)];
                fa_u32x4 w;
                __builtin_memcpy(&w, &v, 16);
                __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, off0 + c * step, 0, 16);
            }
        }
    } else if ((to >> a.lo_sh) < a.Tcur) {
        double *p = a.dst + (i64)dq * a.os_l + FA_TILE_DOFF(a, to);
        const i64 step = 32 * a.os_l;
        const bool sw = (a.flags & FFTW_AMD_F_SWAP_OUT) != 0;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            cplx v = y[slot32(c)];
            if (sw) { double s = v.x; v.x = v.y; v.y = s; }
            *reinterpret_cast<cplx *>(p + c * step) = v;
        }
    }
}

template <bool IN_T, bool OUT_T, int HAS_TW, bool ST_SC1 = false>
FA_DEV void p1024_tile(const P1024Tile &a, double *plane, const int tid) {
    P1024NoHook h;
    p1024_tile<IN_T, OUT_T, HAS_TW, ST_SC1, P1024NoHook>(a, plane, tid, h);
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
};

template <bool IN_T, bool OUT_T, int HAS_TW>
__global__ void __launch_bounds__(256, 2)
pass1024_kernel(const P1024Args a) {
    extern __shared__ __attribute__((aligned(16))) double plane[];
    i64 blk = (i64)blockIdx.x + (i64)blockIdx.y * gridDim.x;
    i64 tile = blk % a.ntiles;
    i64 rest = blk / a.ntiles;
    i64 soff = 0, doff = 0, twb = 0;
    for (int d = 1; d < a.ndims; ++d) {
        i64 idx = rest % a.dn[d];
        rest /= a.dn[d];
        soff += idx * a.dis[d];
        doff += idx * a.dos[d];
        twb += idx * a.dtw[d];
    }
    const i64 t0 = tile * (8 >> a.lo_sh);
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
    p1024_tile<IN_T, OUT_T, HAS_TW>(t, plane, threadIdx.x);
}

#endif /* FA_PASS1024_HPP */
