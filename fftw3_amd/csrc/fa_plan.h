/*
 * fa_plan.h -- internal types of the host planner (C).
 */
#ifndef FA_PLAN_H
#define FA_PLAN_H

#include <stddef.h>
#include <pthread.h>
#include "fftw3.h"
#include "fftw3_amd.h"

typedef long long i64;

#define FA_MAXRANK 8
#define FA_MAXLOOPS 7
#define FA_MAXBUF 16
#define FA_MAXTAB 256
#define FA_MAXPASS 4
#define FA_MAXLANES 4

/* primes up to this get an in-LDS O(p^2) stage; larger ones go to Rader or
   Bluestein (the reference switches from its O(n^2) "generic" solver to
   Rader/Bluestein in the same spirit: fftw/fftw_api.h:1108-1114) */
#define FA_PRIME_LDS_MAX 31

/* complex elements one workgroup tile holds (two LDS images of 16 B each) */
#define FA_TILE_ELEMS 4096
#define FA_LMAX_SINGLE 4096
/* complex elements per LDS image: 160 KiB / (2 images * 16 B) */
#define FA_LDS_ELEMS 5120

enum { FA_C2C = 0, FA_R2C = 1, FA_C2R = 2, FA_R2R = 3 };
#define FA_REAL_IN(t)  ((t) == FA_R2C || (t) == FA_R2R)
#define FA_REAL_OUT(t) ((t) == FA_C2R || (t) == FA_R2R)

enum {
    FA_TAB_STAGE = 1,   /* (cos,sin)(2 pi m / n), m in [0,n) */
    FA_TAB_TW_LO,       /* two-level twiddle, low digits */
    FA_TAB_TW_HI,       /* two-level twiddle, high digits */
    FA_TAB_CHIRP,       /* Bluestein w[k] = (cos,sin)(pi k^2 / n), zero padded to nb */
    FA_TAB_BLUE_SEQ,    /* Bluestein kernel sequence c[k] / nb (time domain) */
    FA_TAB_RADER_SEQ,   /* Rader kernel sequence b[j] / (p-1) (time domain) */
    FA_TAB_DFT_OF,      /* forward DFT of table `src`, computed on the device */
    FA_TAB_PERM         /* int64 index permutation */
};

typedef struct {
    int kind;
    i64 n;          /* defining size */
    i64 aux;        /* shift (TW_*), nb (CHIRP), 0/1 = forward/inverse power (PERM) */
    i64 len;        /* entries: complex pairs, or int64s for PERM */
    int src;        /* FA_TAB_DFT_OF: table id of the time-domain sequence */
    void *host;     /* host copy (NULL for DFT_OF until the device computed it) */
    void *dev;
} fa_table;

typedef struct { i64 n, is, os; } fa_dim;

/* what FFTW_MEASURE tunes and wisdom remembers */
typedef struct {
    size_t chunk_bytes;   /* scratch per chunk */
    int pipeline;         /* two-stream chunk pipeline on/off */
    int lmax_multi;       /* longest sub-transform of a multi-pass split */
    int small_tiles;      /* half-size tiles in the generic LDS kernel */
    int long_first;       /* multi-pass splits run the longest sub-transform first */
    int lanes;            /* chunk lanes: chunk c runs all its steps on stream c % lanes, scratch slot c % lanes */
    i64 tile_elems;       /* FFTW_AMD_TILE_ELEMS: tile size of the generic LDS kernel, 0 = default (not tuned) */
    int real_dec;         /* FFTW_AMD_REAL_DEC: long r2c transforms decimated over the real data (two trips on the 512-item
                             kernels, emit_r2c_decimated) instead of the half- / quarter-length plans; off by default
                             (measured a few % behind them except near n = 2^20), a FFTW_MEASURE candidate */
} fa_cfg;

typedef struct {
    int buf;
    i64 base;
    i64 im;         /* real -> imaginary distance in doubles */
} fa_loc;

typedef struct {
    i64 n;
    i64 is, os;                 /* strides of the transform index, in doubles */
    fa_loc src, dst;
    int nloops;
    fa_dim loops[FA_MAXLOOPS];
    int batch_loop;             /* index of the chunked loop or -1 */
    int flags_in, flags_out;    /* FFTW_AMD_F_* that apply at the first / last step */
    int dense;                  /* the axis interleaved with a 2-element loop is contiguous memory */
} fa_axis;

struct fftw_plan_s {
    int type;                   /* FA_C2C / FA_R2C / FA_C2R / FA_R2R */
    int kinds[FA_MAXRANK];      /* FA_R2R: fftw_r2r_kind of every dim */
    fa_cfg cfg;
    int sign;
    unsigned flags;
    int rank;
    fa_dim dims[FA_MAXRANK];    /* logical transform dims, strides in doubles */
    int hrank;
    fa_dim hdims[FA_MAXRANK];
    i64 in_im, out_im;          /* imaginary offsets of the user arrays */

    fftw_amd_step_desc *steps;
    int nsteps, cap_steps;

    fa_table tabs[FA_MAXTAB];
    int ntabs;

    int nbufs;                  /* ids 0,1 = user in/out */
    i64 buf_reals[FA_MAXBUF];   /* scratch sizes in doubles */
    int buf_busy[FA_MAXBUF];
    double *dbuf[FA_MAXBUF];

    i64 batch, chunk;

    double *ri, *ii, *ro, *io;  /* arrays the plan was created on */
    i64 in_lo, in_hi, out_lo, out_hi;   /* touched offset range of the real parts */
    i64 out_written;            /* number of doubles the plan writes in the output */
    int inplace;
    int single_chunk;           /* run the whole batch as one chunk */
    int via_scratch;            /* in-place c2c problem whose input and output strides differ (same locations): the
                                   result is built in a dense scratch image and copied to the output layout */

    void *stream;
    int dev_ready;
    pthread_mutex_t lock;       /* fftw_execute is thread-safe in the reference; the plan's scratch is not shareable */

    /* chunk pipeline: stage A (steps [0, split)) of chunk c+1 overlaps stage B
       (steps [split, nsteps)) of chunk c on a second stream; each chunk works
       in scratch slot c % nslots */
    int nslots, split;
    int pair;                   /* both passes of the 1024 x 1024 plan in one launch per chunk (fa_hip_launch_pair1024) */
    int lanes;                  /* > 1: chunk lanes (see fa_cfg.lanes); the streams are pstream[0 .. lanes) */
    void *pstream[FA_MAXLANES];
    void *ev_a[4], *ev_b[4], *ev_begin, *ev_end[FA_MAXLANES];
    int failed;

    /* staging for plain host pointers; hstream[0] / [1]: copy streams of the chunked host pipeline */
    void *hstream[2];
    double *stage_in, *stage_out;
    size_t stage_in_bytes, stage_out_bytes;

    double est_flops;

    struct fftw_plan_s *alt;    /* FFTW_UNALIGNED twin, built on the first new-array execution that needs it */

    double *prof_ms;            /* fftw_amd_execute_profiled: per-step sinks, NULL otherwise */
    long long *prof_launches;
};

/* hostmath.c */
void fa_cexp(i64 m, i64 n, double out[2]);
i64  fa_mulmod(i64 x, i64 y, i64 p);
i64  fa_power_mod(i64 b, i64 e, i64 p);
int  fa_prime_factors(i64 n, i64 *primes, int *mult);
int  fa_is_prime(i64 n);
i64  fa_largest_prime_factor(i64 n);
i64  fa_find_generator(i64 p);
int  fa_lds_able(i64 n);
i64  fa_next_smooth(i64 n);
int  fa_radices(i64 L, int *rad);
int  fa_factor_passes(i64 n, int max_passes, i64 lmax_single, i64 lmax_multi, i64 *lens);
int  fa_factor_passes_pref(i64 n, int max_passes, i64 lmax_single, i64 lmax_multi, i64 *lens,
                           int (*tuned)(i64));

/* planner.c */
struct fftw_plan_s *fa_plan_new(void);
void fa_plan_free(struct fftw_plan_s *p);
int  fa_build(struct fftw_plan_s *p);     /* steps from p->type/dims/hdims; 0 on success */
int  fa_device_init(struct fftw_plan_s *p);
void fa_run(struct fftw_plan_s *p, double *ri, double *ii, double *ro, double *io);
char *fa_sprint(const struct fftw_plan_s *p);
fa_cfg fa_default_cfg(void);

/* api.c */
struct fftw_plan_s *fa_unaligned_twin(const struct fftw_plan_s *p);

#endif
