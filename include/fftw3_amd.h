/*
 * fftw3_amd.h -- MI355X-specific additions behind the FFTW3 C ABI.
 *
 * Nothing here exists in the reference ("GPU-specific
 * additions behind the same ABI"); the names live under the fftw_amd_ prefix so
 * the stock fftw_ namespace stays clean.  All signatures are plain C: pointers,
 * sizes and an opaque stream handle (a hipStream_t passed as void*).
 */
#ifndef FFTW3_AMD_EXT_H
#define FFTW3_AMD_EXT_H

#include <stddef.h>
#include "fftw3.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Number of HIP devices visible to the process; 0 when there is none (the
   planners still work, fftw_execute* aborts loudly: there is no CPU fallback). */
int fftw_amd_device_count(void);

/* Device memory (hipMalloc / hipFree).  fftw_execute* accepts these pointers
   and runs on them in place of staging through PCIe.  fftw_malloc() itself
   returns pinned host memory so CPU callers keep working (staged path). */
void *fftw_amd_malloc_device(size_t nbytes);
void  fftw_amd_free_device(void *p);
/* Current device of the calling host thread (hipSetDevice / hipGetDevice) for callers without HIP headers:
   fftw_amd_malloc_device allocates on it.  fftw_amd_set_device returns -1 for a device that does not exist. */
int   fftw_amd_set_device(int device);
int   fftw_amd_get_device(void);
/* Blocking host <-> device copies (hipMemcpy), so that a C caller of the device path needs no
   HIP headers of its own. */
void  fftw_amd_memcpy_to_device(void *dst_device, const void *src_host, size_t nbytes);
void  fftw_amd_memcpy_to_host(void *dst_host, const void *src_device, size_t nbytes);

/* Bind the HIP stream (hipStream_t as void*) that fftw_execute* launches on.
   NULL restores the default stream.  Per plan. */
void fftw_amd_plan_set_stream(fftw_plan p, void *hip_stream);

/* Block until everything launched by fftw_execute*(p) has finished. */
void fftw_amd_plan_sync(fftw_plan p);

/* Bytes of device scratch the plan owns (twiddle tables + work buffers). */
size_t fftw_amd_plan_workspace_bytes(const fftw_plan p);
/* Device that holds them: a plan belongs to the device that was current (fftw_amd_set_device / hipSetDevice)
   when its planner function ran.  -1: none yet / no device; -2: allocations on more than one device (a bug). */
int fftw_amd_plan_workspace_device(const fftw_plan p);

/* Upper bound, in bytes, for the scratch that one chunk of a multi-pass plan may occupy (default
   256 MiB).  Measured on MI355X: with the caller's input and output streamed with nontemporal
   accesses, a 256 MiB scratch image written by one pass is still in the Infinity Cache when the next
   pass reads it (N = 2^20: 12.7 -> 11.4 us per transform); larger chunks lose that, smaller ones pay
   more launches (profiles/r02_chunk_nt_sweep.txt).  FFTW_MEASURE searches
   128 MiB ... 4 GiB.  0 restores the default.  Affects plans created afterwards. */
void fftw_amd_set_chunk_bytes(size_t nbytes);

/* ---- batch sharding over the GPUs of one node ----------------------
   The reference splits the vector loop of a batched plan across its workers inside the library
   (fftw/threads/dft-vrank-geq1.c:140-175, block size ceil(vl / nthr) :158-159; the same block
   rule as fftw/mpi/block.c:35-42).  Here the workers are GPUs: shard g of ndev owns the
   transforms [g*ceil(B/ndev), min(B, (g+1)*ceil(B/ndev))) of the batch, with one plan replica
   (tables + scratch), one stream and one host thread per device, and nothing exchanged inside a
   transform.  in[g] / out[g] are device pointers on device devs[g] (devs == NULL: 0..ndev-1)
   that hold shard g -- its first transform at offset 0 -- laid out exactly as
   fftw_plan_many_dft describes one batch (embed, stride, dist).  A device may appear more than
   once in devs (its shards then run on separate streams of that device). */
typedef struct fftw_amd_sharded_plan_s *fftw_amd_sharded_plan;

void fftw_amd_shard_range(long long howmany, int nshards, int g, long long *lo, long long *hi);

fftw_amd_sharded_plan fftw_amd_plan_many_dft_sharded(
    int rank, const int *n, int howmany, int ndev, const int *devs,
    fftw_complex *const *in, const int *inembed, int istride, int idist,
    fftw_complex *const *out, const int *onembed, int ostride, int odist, int sign, unsigned flags);
fftw_amd_sharded_plan fftw_amd_plan_many_dft_r2c_sharded(
    int rank, const int *n, int howmany, int ndev, const int *devs,
    double *const *in, const int *inembed, int istride, int idist,
    fftw_complex *const *out, const int *onembed, int ostride, int odist, unsigned flags);
fftw_amd_sharded_plan fftw_amd_plan_many_dft_c2r_sharded(
    int rank, const int *n, int howmany, int ndev, const int *devs,
    fftw_complex *const *in, const int *inembed, int istride, int idist,
    double *const *out, const int *onembed, int ostride, int odist, unsigned flags);

/* Enqueue every shard's transforms (asynchronous, like fftw_execute on device pointers);
   fftw_amd_sharded_sync blocks until all devices are done. */
void fftw_amd_execute_sharded(const fftw_amd_sharded_plan p);
void fftw_amd_sharded_sync(const fftw_amd_sharded_plan p);

/* All-gather of the output shards over xGMI: full[d] is a buffer on shard d's device for the
   WHOLE batch (howmany * odist elements); every shard's output lands at its place in each of
   them, ordered after the transforms (complete after fftw_amd_sharded_sync).  mode 0: RCCL
   (librccl.so, loaded once per process; ncclAllGather when the shards are equal, one ncclBroadcast per
   shard in a group when the block rule leaves a ragged tail; issued on the shards' own streams behind
   their transforms, no host-side drain) when every shard has its own device, direct peer-to-peer pushes
   otherwise; 1: force peer-to-peer; 2: RCCL or fail.  Returns 1 if RCCL moved the data, 0 for
   peer-to-peer, -1 on error.  This is the xGMI-bound part (for cfg5 each GPU
   receives 120 GB at <= 7 x 153 GB/s) and is never part of a transform's timing. */
int fftw_amd_sharded_all_gather(const fftw_amd_sharded_plan p, void *const *full, int mode);
/* The RCCL calls fftw_amd_sharded_all_gather issues (inside one group), without issuing them: 6 values per
   call -- kind (0 ncclAllGather, 1 ncclBroadcast), rank d, root (-1 for all-gather), send pointer, receive
   pointer, bytes.  Equal shards: one ncclAllGather per rank; ragged block-rule tails: one ncclBroadcast per
   non-empty shard and rank.  Returns the number of calls (ops holds at most cap of them), -1 on bad arguments. */
int fftw_amd_sharded_gather_ops(const fftw_amd_sharded_plan p, void *const *full, long long *ops, int cap);
/* Number of RCCL entry points the loader resolves (7 when librccl.so can serve the gather, else 0); opens the
   library once per process, makes no RCCL call.  FFTW_AMD_RCCL_LIB names a stand-in library. */
int fftw_amd_rccl_probe(void);

int  fftw_amd_sharded_num_shards(const fftw_amd_sharded_plan p);
int  fftw_amd_sharded_device(const fftw_amd_sharded_plan p, int g);
void fftw_amd_sharded_range(const fftw_amd_sharded_plan p, int g, long long *lo, long long *hi);
fftw_plan fftw_amd_sharded_replica(const fftw_amd_sharded_plan p, int g);   /* NULL for an empty shard */
void fftw_amd_destroy_sharded_plan(fftw_amd_sharded_plan p);

/* ---- one transform spread over the GPUs of a node in slabs ----------------------
   The c2c part of the reference's distributed-memory API (fftw/mpi/fftw3-mpi.h:74-215: fftw_mpi_local_size_2d / _3d,
   fftw_mpi_plan_dft_2d / _3d, fftw_mpi_execute_dft) with the MPI communicator replaced by a list of devices of this
   process: device g of ndev owns the rows [local_0_start, local_0_start + local_n0) of the first dimension (block
   rule of fftw/mpi/block.c:39-50), normal order in and out.  Pipeline: local transforms over the trailing
   dimension(s), exchange of column blocks (peer-to-peer 2-D copies over xGMI on the receiver's stream), local
   transforms of length n0 down the column blocks, exchange back (fftw3_amd/csrc/slab.c).  The multi-process form of
   the same layer -- r2c / c2r / r2r, TRANSPOSED_IN / OUT, torch.distributed in the place of MPI -- is
   fftw3_amd/slab.py. */
typedef struct fftw_amd_slab_plan_s *fftw_amd_slab_plan;
/* rows of device g (rank = 2 or 3, n = the logical size); returns the number of complex elements of its arrays */
long long fftw_amd_slab_local_size(int rank, const long long *n, int ndev, int g, long long *local_n0, long long *local_0_start);
/* in[g] / out[g]: device arrays on devs[g] (devs == NULL: 0..ndev-1) with local_n0(g) x n[1] (x n[2]) complex values;
   in[g] == out[g] is allowed.  NULL on invalid arguments or when a local plan / buffer cannot be made. */
fftw_amd_slab_plan fftw_amd_slab_plan_dft(int rank, const long long *n, int ndev, const int *devs,
                                          fftw_complex *const *in, fftw_complex *const *out, int sign, unsigned flags);
void fftw_amd_slab_execute(fftw_amd_slab_plan p);      /* enqueues on every device and returns */
void fftw_amd_slab_sync(fftw_amd_slab_plan p);
int  fftw_amd_slab_num_devices(const fftw_amd_slab_plan p);
fftw_plan fftw_amd_slab_local_plan(const fftw_amd_slab_plan p, int g, int which);   /* 0: rows plan, 1: column-block plan */
void fftw_amd_destroy_slab_plan(fftw_amd_slab_plan p);

/* ---- plan introspection used by the host-logic tests ------------------- */

#define FFTW_AMD_MAX_DIMS 8
#define FFTW_AMD_MAX_RADICES 16

/* One launch of the executor, as the planner built it.  Offsets and strides
   are counted in doubles (an interleaved complex array has stride 2, im = 1). */
typedef struct {
    int kind;            /* FFTW_AMD_STEP_* */
    int src_buf, dst_buf;/* 0 = plan input, 1 = plan output, 2.. = scratch */
    long long src_base, dst_base;
    long long src_im, dst_im;     /* distance from real to imaginary part */
    int flags;                    /* FFTW_AMD_F_* */
    int L;                        /* sub-transform length of a PASS */
    int nradices, radices[FFTW_AMD_MAX_RADICES];
    long long is_l, os_l;         /* stride of the transform index */
    int ndims;                    /* dims[0] is the tile dim, the rest loops */
    long long dim_n[FFTW_AMD_MAX_DIMS];
    long long dim_is[FFTW_AMD_MAX_DIMS];
    long long dim_os[FFTW_AMD_MAX_DIMS];
    long long dim_tw[FFTW_AMD_MAX_DIMS];
    long long tw_n;               /* 0: no inter-pass twiddle */
    int tw_shift, tw_lo, tw_hi;   /* two-level table: w^m = lo[m & mask] * hi[m >> shift] */
    int tile;                     /* tile width T along dims[0] */
    int batch_dim;                /* index into dims of the chunked batch loop, or -1 */
    long long aux_n;              /* kind-specific length (r2c: n, copy: K ...) */
    long long aux_valid;          /* copy: source elements beyond this read as 0 */
    int table, table2;            /* stage-twiddle / multiplier table, permutation table; -1: none */
    int aux_buf;                  /* Rader: buffer holding x[0] per vector */
    long long aux_base;
    int variant;                  /* which kernel the executor launches (FFTW_AMD_K_*) */
    /* optional inner component of the tile dim: the sequences of a tile are
       (lo, hi) pairs, lo in [0, tile_lo_n) fastest; dims[0] describes hi.
       Semantically just one more loop; it lets two interleaved vectors (radix-4
       real transforms) share 16-byte-contiguous global accesses. */
    int tile_lo_n;
    long long tile_lo_is, tile_lo_os;
    /* element-wise real-transform steps (untangle / tangle / r2r): the work items run over
       dims[0..kpos), then the pair / transform index, then the remaining dims -- the planner
       orders them by the user-side stride so that neighbouring items touch neighbouring
       memory whichever axis is transformed */
    int kpos;
} fftw_amd_step_desc;

enum {
    FFTW_AMD_STEP_PASS = 1,     /* batched length-L DFT + optional twiddle */
    FFTW_AMD_STEP_COPY = 2,     /* strided copy / pad / table multiply / permute */
    FFTW_AMD_STEP_R2C_POST = 3, /* untangle half-length complex DFT into r2c output */
    FFTW_AMD_STEP_C2R_PRE = 4,  /* inverse of the above */
    FFTW_AMD_STEP_RADER_MUL = 5,/* Rader pointwise product and DC fix-ups */
    FFTW_AMD_STEP_HERM_EXPAND = 6,/* half spectrum -> full Hermitian spectrum */
    FFTW_AMD_STEP_R2C_POST4 = 7,  /* radix-4 untangle: two quarter-length complex DFTs -> r2c output */
    FFTW_AMD_STEP_C2R_PRE4 = 8,   /* inverse of the above */
    FFTW_AMD_STEP_R2R = 9         /* r2r pre/post-processing around a real or complex DFT; variant = FFTW_AMD_R2R_* */
};

/* FFTW_AMD_STEP_R2R modes (step.variant).  aux_n = r2r length n, aux_valid =
   work items per transform, kpos = position of the transform index among the
   flattened loop indices, tw_* = two-level table of modulus 4n (01/10) or 8n (11).
   PRE_* map the user's real array into the inner transform's input, POST_* map
   the inner transform's output into the user's real array. */
enum {
    FFTW_AMD_R2R_PRE_HC2R = 1,   /* halfcomplex -> interleaved half spectrum */
    FFTW_AMD_R2R_PRE_E10, FFTW_AMD_R2R_PRE_O10,     /* even/odd index shuffle (sign flips for RODFT10) */
    FFTW_AMD_R2R_PRE_E01, FFTW_AMD_R2R_PRE_O01,     /* twiddled half spectrum for c2r */
    FFTW_AMD_R2R_PRE_E00, FFTW_AMD_R2R_PRE_O00,     /* even / odd symmetric extension */
    FFTW_AMD_R2R_PRE_E11, FFTW_AMD_R2R_PRE_O11,     /* n even: n/2 twiddled complex pairs */
    FFTW_AMD_R2R_PRE_E11ODD, FFTW_AMD_R2R_PRE_O11ODD, /* n odd: twiddle and zero-pad to 2n */
    FFTW_AMD_R2R_POST_R2HC, FFTW_AMD_R2R_POST_DHT,
    FFTW_AMD_R2R_POST_E10, FFTW_AMD_R2R_POST_O10,
    FFTW_AMD_R2R_POST_E01, FFTW_AMD_R2R_POST_O01,
    FFTW_AMD_R2R_POST_E00, FFTW_AMD_R2R_POST_O00,
    FFTW_AMD_R2R_POST_E11, FFTW_AMD_R2R_POST_O11,
    FFTW_AMD_R2R_POST_E11ODD, FFTW_AMD_R2R_POST_O11ODD
};

enum {
    FFTW_AMD_K_GENERIC = 0,     /* runtime-radix LDS kernel */
    FFTW_AMD_K_P1024 = 1,       /* register-resident radix-32x32 kernel, tile of 8 */
    FFTW_AMD_K_RR = 2,          /* register-resident two-stage kernel, L = 64..512, tile of 8192/L */
    FFTW_AMD_K_R3 = 3,          /* register-resident three-stage kernels (rows up to 8192 and of 16384 points; strided up to 2048) */
    FFTW_AMD_K_R2C = 4,         /* fused real rows -> half spectra (r2crows.hpp), with FFTW_AMD_F_R2C_ROWS */
    FFTW_AMD_K_C2R = 5,         /* fused half spectra -> real rows, with FFTW_AMD_F_C2R_ROWS */
    FFTW_AMD_K_R1 = 6,          /* one-stage register kernel: dense rows of 2 ... 32 points, one butterfly per row */
    FFTW_AMD_K_BLUE = 7         /* Bluestein's algorithm for a whole row in one kernel (pass3b.hpp): L = padded length nb,
                                   aux_n = n, tw_lo / tw_hi = ids of the chirp and the kernel table */
};

enum {
    FFTW_AMD_F_SWAP_IN   = 1 << 0, /* read (im,re) instead of (re,im) */
    FFTW_AMD_F_SWAP_OUT  = 1 << 1, /* write (im,re) */
    FFTW_AMD_F_REAL_IN   = 1 << 2, /* source has no imaginary part (reads as 0) */
    FFTW_AMD_F_REAL_OUT  = 1 << 3, /* store the real part only */
    FFTW_AMD_F_MUL_TABLE = 1 << 4, /* copy: multiply by table[k] */
    FFTW_AMD_F_MUL_CONJ  = 1 << 5, /* copy: multiply by conj(table[k]) */
    FFTW_AMD_F_PERM_SRC  = 1 << 6, /* copy: source index through permutation */
    FFTW_AMD_F_PERM_DST  = 1 << 7, /* copy: destination index through permutation */
    FFTW_AMD_F_CONJ_OUT  = 1 << 8, /* conjugate on store */
    FFTW_AMD_F_TW_IN     = 1 << 9, /* pass: the twiddle multiplies the INPUT element (l, q) instead of the output */
    FFTW_AMD_F_R2C_ROWS  = 1 << 10,/* pass over real pairs (src_im = 1) that also does the r2c untangle for n = 2L:
                                      stores L + 1 entries per row; tw_lo / tw_hi hold w_n^m (tw_n stays 0) */
    FFTW_AMD_F_C2R_ROWS  = 1 << 11,/* the transpose: reads L + 1 spectrum entries per row, c2r tangle, backward
                                      length-L pass, stores the real pairs (dst_im = 1) */
    FFTW_AMD_F_NT_IN     = 1 << 12,/* the source is read once per execution: nontemporal loads */
    FFTW_AMD_F_NT_OUT    = 1 << 13,/* the destination is not read again by this plan: nontemporal stores */
    FFTW_AMD_F_LO_DFT    = 1 << 14,/* pass: the inner tile component is transformed too -- a DFT of length tile_lo_n across the
                                      tile's tile_lo_n sequences (no twiddle), i.e. the step is the 2-D DFT tile_lo_n x L of every
                                      tile (pass3q.hpp: four rows of 4096 points, the last trip of a 4096 x 4096 transform) */
    FFTW_AMD_F_REAL_DEC  = 1 << 15 /* pass: the last trip of a two-trip r2c transform of n = L1 x L real points (pass3t_kernel,
                                      RD = 1).  The tile dim runs over the rows k1 = 0 ... L1 / 2 (dim_n[0] = L1 / 2 + 1) of the
                                      scratch image Z[k1][c] (rows at dim_is[0], pairs at is_l) left by the complex pass of
                                      length L1 over the real input read as pairs; row k1 is loaded as
                                      A[2c] = (Z[k1][c] + conj Z[L1-k1][c]) / 2, A[2c+1] = (Z[k1][c] - conj Z[L1-k1][c]) / 2i,
                                      multiplied by the input twiddle and transformed; output k2 < L / 2 goes to
                                      k1 dim_os[0] + k2 os_l, the others conjugated to (L1-k1) dim_os[0] + (L-1-k2) os_l
                                      (not for k1 = 0 and L1 / 2, whose second halves repeat their first; X[n / 2] comes from
                                      row 0).  No fallback executor: planned only for aligned interleaved arrays */
};

int fftw_amd_plan_num_steps(const fftw_plan p);
int fftw_amd_plan_get_step(const fftw_plan p, int i, fftw_amd_step_desc *out);
/* how many batch elements one chunk processes, and the total batch */
long long fftw_amd_plan_chunk(const fftw_plan p);
long long fftw_amd_plan_batch(const fftw_plan p);
/* 1 when the executor runs both passes of a two-pass plan in one launch per chunk (pass 2 of chunk
   c-1 followed by pass 1 of chunk c); fftw_amd_execute_profiled then reports every launch under step 0.
   Known once the plan has been set up on a device. */
int fftw_amd_plan_paired(const fftw_plan p);
/* Chunk lanes of a multi-pass plan (round 3): chunk c of the batch runs ALL its steps, in order, on stream
   c % lanes in scratch slot c % lanes; the lanes share the chip, so the tail of one lane's launch is filled by
   the other lane's next one and a chunk's scratch is re-read while it is still in the Infinity Cache
   (default 2 lanes x 128 MiB of scratch; FFTW_AMD_LANES=1 restores serial chunks, and with them the pair
   launches above).  The lanes fork from and join the plan's stream (fftw_amd_plan_set_stream) inside every
   fftw_execute*.  1 for single-chunk, one-step and oversized-chunk plans.  Known once the plan is set up. */
int fftw_amd_plan_lanes(const fftw_plan p);
/* host copy of table `id` as interleaved doubles; returns its length in
   doubles (writes at most cap doubles). */
long long fftw_amd_plan_table(const fftw_plan p, int id, double *dst, long long cap);

/* Execute once on the plan's own arrays with a HIP event pair around every
   kernel launch, on the stream the kernels run on.  ms[i] receives the summed
   duration of step i over all chunks, launches[i] how often it was launched.
   Returns the number of steps, -1 if cap is too small. */
int fftw_amd_execute_profiled(fftw_plan p, double *ms, long long *launches, int cap);

/* Host-side numerics exported for the tests (no device needed). */
void fftw_amd_cexp(long long m, long long n, double out[2]);  /* (cos, sin)(2 pi m / n) */
long long fftw_amd_find_generator(long long p);
long long fftw_amd_power_mod(long long b, long long e, long long p);
int fftw_amd_factor_passes(long long n, int max_passes, long long *lens);

#ifdef __cplusplus
}
#endif
#endif
