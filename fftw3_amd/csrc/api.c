/*
 * api.c -- the FFTW3 C API on top of the GPU planner.
 *
 * Argument meaning, defaults and error behaviour follow the reference API
 * layer (fftw/fftw_api.c:1-1516); citations per function.  Errors are
 * reported the FFTW way: planners return NULL, internal failures abort().
 */
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "fa_plan.h"
#include "fa_hip.h"

typedef struct fftw_plan_s plan;

const char fftw_version[] = "fftw3-amd-0.1 (MI355X/gfx950 HIP executor, FFTW 3.3 API)";
const char fftw_cc[] = "hipcc --offload-arch=gfx950";
const char fftw_codelet_optim[] = "";


/* ------------------------------------------------------------- wisdom */
/* What the reference's planner learns by timing (ifftw_mkplan + wisdom hash
   table, fftw/fftw_api.c:15300-15426, 14829-14838) becomes here: which launch
   configuration is fastest for a problem -- scratch chunk size, the two-stream
   chunk pipeline, the longest sub-transform of a multi-pass split, generic
   kernel tile size.  FFTW_ESTIMATE takes the static defaults; any other flag
   times the candidates on the device (overwriting the arrays, as FFTW_MEASURE
   does in the reference: A.c:18604) and records the winner as wisdom, which the
   wisdom export / import calls move between processes as text. */
/* room for 2 * FA_MAXRANK dims of three 20-digit numbers each, plus the fixed fields */
#define FA_WIS_KEY_CAP 1280
typedef struct wis_s {
    struct wis_s *next;
    char key[FA_WIS_KEY_CAP];
    fa_cfg cfg;
    double ms;
} wis_entry;
static wis_entry *g_wisdom = NULL;

/* bounded append: never writes past key[cap - 1], however long the text would have been
   (snprintf returns the would-be length, which must not be trusted as the new end) */
static void kapp(char *key, size_t cap, size_t *len, const char *fmt, ...) {
    va_list ap;
    int n;
    if (*len + 1 >= cap) return;
    va_start(ap, fmt);
    n = vsnprintf(key + *len, cap - *len, fmt, ap);
    va_end(ap);
    if (n < 0) return;
    *len += (size_t)n;
    if (*len > cap - 1) *len = cap - 1;
}

static void wis_key(const plan *p, char *key, size_t cap) {
    size_t len = 0;
    int i;
    key[0] = 0;
    kapp(key, cap, &len, "t%d s%d r%d", p->type, p->sign, p->rank);
    for (i = 0; i < p->rank; ++i)
        kapp(key, cap, &len, " %lld:%lld:%lld", p->dims[i].n, p->dims[i].is, p->dims[i].os);
    if (p->type == FA_R2R) {
        kapp(key, cap, &len, " k");
        for (i = 0; i < p->rank; ++i) kapp(key, cap, &len, "%d.", p->kinds[i]);
    }
    kapp(key, cap, &len, " h%d", p->hrank);
    for (i = 0; i < p->hrank; ++i)
        kapp(key, cap, &len, " %lld:%lld:%lld", p->hdims[i].n, p->hdims[i].is, p->hdims[i].os);
    kapp(key, cap, &len, " i%lld o%lld p%d", p->in_im, p->out_im, p->inplace);
}

static wis_entry *wis_find(const char *key) {
    wis_entry *w;
    for (w = g_wisdom; w; w = w->next) if (!strcmp(w->key, key)) return w;
    return NULL;
}

static void wis_put(const char *key, fa_cfg cfg, double ms) {
    wis_entry *w = wis_find(key);
    if (!w) {
        w = (wis_entry *)calloc(1, sizeof(*w));
        if (!w) return;
        snprintf(w->key, sizeof(w->key), "%s", key);
        w->next = g_wisdom;
        g_wisdom = w;
    }
    w->cfg = cfg;
    w->ms = ms;
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

/* copy the problem description of `p` into a fresh plan with configuration c */
static plan *clone_problem(const plan *p, fa_cfg c) {
    plan *q = fa_plan_new();
    int i;
    if (!q) return NULL;
    q->type = p->type; q->sign = p->sign; q->flags = p->flags;
    q->rank = p->rank; q->hrank = p->hrank;
    for (i = 0; i < p->rank; ++i) { q->dims[i] = p->dims[i]; q->kinds[i] = p->kinds[i]; }
    for (i = 0; i < p->hrank; ++i) q->hdims[i] = p->hdims[i];
    q->in_im = p->in_im; q->out_im = p->out_im;
    q->single_chunk = p->single_chunk;
    q->via_scratch = p->via_scratch;
    q->cfg = c;
    return q;
}

/* The twin of a plan that serves arrays of ANY alignment (built with FFTW_UNALIGNED, so the planner keeps to
   kernels that have an executor for every layout): what fa_run switches to when a new-array execution passes
   arrays aligned differently from the ones the plan was created on.  FFTW's contract makes that the caller's
   error (fftw3.h, new-array execute functions; A.c:433-440 does not check); the one-trip rows kernels used to
   abort() the process on it.  NULL if the twin cannot be built. */
plan *fa_unaligned_twin(const plan *p) {
    plan *q = clone_problem(p, p->cfg);
    if (!q) return NULL;
    q->flags = (p->flags | FFTW_UNALIGNED | FFTW_ESTIMATE) & ~(unsigned)(FFTW_MEASURE | FFTW_PATIENT | FFTW_EXHAUSTIVE);
    q->ri = p->ri; q->ii = p->ii; q->ro = p->ro; q->io = p->io;
    q->inplace = p->inplace;
    if (fa_build(q)) { fa_plan_free(q); return NULL; }
    q->in_lo = p->in_lo; q->in_hi = p->in_hi; q->out_lo = p->out_lo; q->out_hi = p->out_hi;
    q->out_written = p->out_written;
    return q;
}

/* ------------------------------------------------------------ helpers */

/* offset range touched by a strided tensor, relative to its base */
static void span_of(const fa_dim *d, int nd, int use_os, i64 *lo, i64 *hi) {
    int i;
    for (i = 0; i < nd; ++i) {
        i64 s = use_os ? d[i].os : d[i].is;
        i64 e = (d[i].n - 1) * s;
        if (d[i].n <= 0) continue;
        if (e < 0) *lo += e; else *hi += e;
    }
}

static plan *finish_locked(plan *p, double *ri, double *ii, double *ro, double *io) {
    i64 cnt = 1;
    int i;
    p->ri = ri; p->ii = ii; p->ro = ro; p->io = io;
    p->inplace = (ri == ro);
    p->in_lo = p->in_hi = p->out_lo = p->out_hi = 0;
    span_of(p->dims, p->rank, 0, &p->in_lo, &p->in_hi);
    span_of(p->hdims, p->hrank, 0, &p->in_lo, &p->in_hi);
    span_of(p->dims, p->rank, 1, &p->out_lo, &p->out_hi);
    span_of(p->hdims, p->hrank, 1, &p->out_lo, &p->out_hi);
    if (p->type == FA_R2C && p->rank > 0) {
        /* the complex side has n/2+1 entries along the last dim */
        i64 nl = p->dims[p->rank - 1].n, s = p->dims[p->rank - 1].os;
        i64 full = (nl - 1) * s, half = (nl / 2) * s;
        if (s >= 0) p->out_hi += half - full; else p->out_lo += half - full;
    }
    if (p->type == FA_C2R && p->rank > 0) {
        i64 nl = p->dims[p->rank - 1].n, s = p->dims[p->rank - 1].is;
        i64 full = (nl - 1) * s, half = (nl / 2) * s;
        if (s >= 0) p->in_hi += half - full; else p->in_lo += half - full;
    }
    for (i = 0; i < p->rank; ++i) {
        i64 e = p->dims[i].n;
        if (p->type == FA_R2C && i == p->rank - 1) e = e / 2 + 1;
        cnt *= e;
    }
    for (i = 0; i < p->hrank; ++i) cnt *= p->hdims[i].n;
    p->out_written = FA_REAL_OUT(p->type) ? cnt : 2 * cnt;
    {
        char key[FA_WIS_KEY_CAP];
        wis_entry *w;
        wis_key(p, key, sizeof(key));
        w = wis_find(key);
        if (w) {
            p->cfg = w->cfg;
        } else if (p->flags & FFTW_WISDOM_ONLY) {
            fa_plan_free(p);               /* reference: NULL when no wisdom applies (fftw3.h FFTW_WISDOM_ONLY) */
            return NULL;
        } else if (!(p->flags & FFTW_ESTIMATE) && ri && ro && fa_hip_device_count() > 0) {
            /* FFTW_MEASURE / PATIENT / EXHAUSTIVE: time candidate configurations on the device */
            static const size_t chunks[] = { (size_t)64 << 20, (size_t)128 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)4 << 30 };
            fa_cfg best = p->cfg, c;
            double best_ms = -1.0;
            int ci, pi, li, st, lf, rd, nl = (p->flags & (FFTW_PATIENT | FFTW_EXHAUSTIVE)) ? 2 : 1;
            /* r2c problems: also the plan decimated over the real data (cfg.real_dec), with the default chunking only */
            const int nrd = (p->type == FA_R2C) ? 2 : 1;
            for (rd = 0; rd < nrd; ++rd)
            for (ci = 0; ci < 5; ++ci)
                for (pi = 0; pi < 2; ++pi)
                    for (li = 0; li < nl; ++li)
                        for (st = 0; st < nl; ++st)
                        for (lf = 0; lf < 2; ++lf) {
                            if (rd && (ci != 1 || pi || li || st || lf)) continue;
                            plan *q;
                            double t0, dt;
                            c = p->cfg;
                            c.chunk_bytes = chunks[ci];
                            c.pipeline = pi;
                            c.lanes = pi ? 1 : (ci < 3 ? 2 : 1);   /* small chunks: two chunk lanes; large ones: serial or the stage pipeline */
                            c.lmax_multi = li ? 512 : 1024;
                            c.small_tiles = st;
                            c.long_first = lf;
                            c.real_dec = rd;
                            q = clone_problem(p, c);
                            if (!q) continue;
                            /* the builder looks at the arrays (alignment, aliasing) */
                            q->ri = ri; q->ii = ii; q->ro = ro; q->io = io;
                            q->inplace = p->inplace;
                            if (fa_build(q)) { fa_plan_free(q); continue; }
                            q->in_lo = p->in_lo; q->in_hi = p->in_hi; q->out_lo = p->out_lo; q->out_hi = p->out_hi;
                            q->out_written = p->out_written;
                            fa_run(q, ri, ii, ro, io);          /* warm-up, device init */
                            fa_hip_stream_sync(q->stream);
                            t0 = now_ms();
                            fa_run(q, ri, ii, ro, io);
                            fa_run(q, ri, ii, ro, io);
                            fa_hip_stream_sync(q->stream);
                            dt = 0.5 * (now_ms() - t0);
                            if (best_ms < 0 || dt < best_ms) { best_ms = dt; best = c; }
                            fa_plan_free(q);
                        }
            p->cfg = best;
            wis_put(key, best, best_ms);
        }
    }
    if (fa_build(p)) { fa_plan_free(p); return NULL; }
    /* tables and scratch are set up now when a device is present, so that
       fftw_execute itself allocates nothing (stream capture, latency) */
    if (fa_hip_device_count() > 0 && p->batch > 0 && fa_device_init(p)) { fa_plan_free(p); return NULL; }
    return p;
}

/* The reference's planner is not reentrant unless fftw_make_planner_thread_safe() was
   called (fftw/fftw_api.c:14517-14540, threads hook); here the planner's shared state is
   the wisdom list and the device bring-up, and one lock around them is cheap enough to
   take always.  Executing different plans from different threads needs no lock at all. */
static pthread_mutex_t g_planner_lock = PTHREAD_MUTEX_INITIALIZER;

static plan *finish(plan *p, double *ri, double *ii, double *ro, double *io) {
    plan *r;
    pthread_mutex_lock(&g_planner_lock);
    r = finish_locked(p, ri, ii, ro, io);
    pthread_mutex_unlock(&g_planner_lock);
    return r;
}

/* in-place transforms must map every element onto itself
   (reference fftw_mkproblem_dft_d rejects the rest: fftw/fftw_api.c:4090-4094) */
static int inplace_ok_c2c(const plan *p) {
    int i;
    for (i = 0; i < p->rank; ++i) if (p->dims[i].is != p->dims[i].os) return 0;
    for (i = 0; i < p->hrank; ++i) if (p->hdims[i].is != p->hdims[i].os) return 0;
    return 1;
}

/* In-place problems whose input and output strides differ are still legal in the reference when both stride sets
   address the SAME locations (fftw_tensor_inplace_locations, A.c:17298-17311: the transform and vector dims, taken
   once with the input and once with the output strides, compress to equal tensors) -- in-place transposes and
   transposed-output transforms, which the reference plans with its square DIF + transpose codelets
   (dft-ct-dif + dftw-directsq "q1_r", A.c:2204-2250, 2596-2727) or buffers.  Same test here: sort the (n, |stride|)
   pairs of each side by stride, merge contiguous runs, compare. */
static int locs_compress(const plan *p, int use_os, i64 *n, i64 *st) {
    int i, j, k = 0;
    for (i = 0; i < p->rank + p->hrank; ++i) {
        const fa_dim *d = i < p->rank ? &p->dims[i] : &p->hdims[i - p->rank];
        i64 s = use_os ? d->os : d->is;
        if (d->n == 1) continue;
        n[k] = d->n; st[k] = s < 0 ? -s : s; ++k;
    }
    for (i = 1; i < k; ++i)                       /* insertion sort by stride */
        for (j = i; j > 0 && st[j] < st[j - 1]; --j) {
            i64 t = st[j]; st[j] = st[j - 1]; st[j - 1] = t;
            t = n[j]; n[j] = n[j - 1]; n[j - 1] = t;
        }
    for (i = 0, j = 0; i < k; ++i) {              /* merge: next stride == n * stride */
        if (j > 0 && st[i] == n[j - 1] * st[j - 1]) n[j - 1] *= n[i];
        else { n[j] = n[i]; st[j] = st[i]; ++j; }
    }
    return j;
}
static int inplace_same_locations(const plan *p) {
    i64 ni[2 * FA_MAXRANK], si[2 * FA_MAXRANK], no[2 * FA_MAXRANK], so[2 * FA_MAXRANK];
    int i, ki, ko;
    for (i = 0; i < p->rank + p->hrank; ++i) {
        const fa_dim *d = i < p->rank ? &p->dims[i] : &p->hdims[i - p->rank];
        if (d->n > 1 && (d->is < 0 || d->os < 0)) return 0;      /* mirrored layouts: not taken on */
    }
    ki = locs_compress(p, 0, ni, si);
    ko = locs_compress(p, 1, no, so);
    if (ki != ko) return 0;
    for (i = 0; i < ki; ++i) if (ni[i] != no[i] || si[i] != so[i]) return 0;
    return 1;
}

/* move the loop with the largest stride to hdims[0]: it is the chunked batch */
static void pick_batch(plan *p) {
    int i, best = 0;
    fa_dim t;
    if (p->hrank < 2) return;
    for (i = 1; i < p->hrank; ++i) {
        i64 a = p->hdims[i].is < 0 ? -p->hdims[i].is : p->hdims[i].is;
        i64 b = p->hdims[best].is < 0 ? -p->hdims[best].is : p->hdims[best].is;
        if (a > b) best = i;
    }
    t = p->hdims[0]; p->hdims[0] = p->hdims[best]; p->hdims[best] = t;
}

static int dims_ok(int rank, const fftw_iodim64 *d) {
    int i;
    if (rank < 0 || rank > FA_MAXRANK) return 0;
    for (i = 0; i < rank; ++i) if (d[i].n <= 0) return 0;
    return 1;
}

static int hdims_ok(int rank, const fftw_iodim64 *d) {
    int i;
    if (rank < 0 || rank > FA_MAXRANK) return 0;
    for (i = 0; i < rank; ++i) if (d[i].n < 0) return 0;
    return 1;
}

/* common constructor: dims/hdims strides in units of the element type of each
   side; mul_i / mul_o convert them to doubles (2 for complex, 1 for real) */
static plan *mk_guru(int type, int rank, const fftw_iodim64 *dims, int hrank,
                     const fftw_iodim64 *hdims, double *ri, double *ii, double *ro, double *io,
                     int mul_i, int mul_o, int sign, unsigned flags) {
    plan *p;
    int i, k = 0;
    if (!dims_ok(rank, dims) || !hdims_ok(hrank, hdims)) return NULL;
    if ((type != FA_C2C) && rank < 1) return NULL;
    p = fa_plan_new();
    if (!p) return NULL;
    p->type = type;
    p->sign = sign;
    p->flags = flags;
    p->rank = rank;
    for (i = 0; i < rank; ++i) {
        p->dims[i].n = dims[i].n;
        p->dims[i].is = dims[i].is * mul_i;
        p->dims[i].os = dims[i].os * mul_o;
    }
    for (i = 0; i < hrank; ++i) {
        if (hdims[i].n == 1) continue;          /* extent-1 loops carry nothing */
        p->hdims[k].n = hdims[i].n;
        p->hdims[k].is = hdims[i].is * mul_i;
        p->hdims[k].os = hdims[i].os * mul_o;
        ++k;
    }
    p->hrank = k;
    pick_batch(p);
    p->in_im = (type == FA_R2C) ? 0 : (i64)(ii - ri);
    p->out_im = (type == FA_C2R) ? 0 : (i64)(io - ro);
    if (type == FA_C2C && ri == ro && !inplace_ok_c2c(p)) {
        /* strides differ: legal when both sides address the same locations; the whole problem then goes through a
           dense scratch image (every read before every write), in one chunk */
        if (ii != io || !inplace_same_locations(p)) { fa_plan_free(p); return NULL; }
        p->via_scratch = 1;
        p->single_chunk = 1;
    }
    if (type != FA_C2C && (void *)ri == (void *)ro) {
        /* in-place real transforms always pass through scratch; when the two
           layouts do not advance together per batch element, run the whole
           batch as one chunk so every read precedes every write */
        int same = 1;
        for (i = 0; i < p->hrank; ++i) if (p->hdims[i].is != p->hdims[i].os) same = 0;
        p->single_chunk = !same;
        return finish(p, ri, ii, ro, io);
    }
    return finish(p, ri, ii, ro, io);
}

static void to64(int rank, const fftw_iodim *d, fftw_iodim64 *o) {
    int i;
    for (i = 0; i < rank && i < FA_MAXRANK; ++i) { o[i].n = d[i].n; o[i].is = d[i].is; o[i].os = d[i].os; }
}

/* row-major dims with physical (embedding) sizes -> strides
   (reference fftw_mktensor_rowmajor, fftw/fftw_api.c:842-861) */
static void rowmajor(int rank, const int *n, const int *niphys, const int *nophys,
                     i64 is, i64 os, fftw_iodim64 *d) {
    int i;
    if (rank <= 0) return;
    d[rank - 1].n = n[rank - 1];
    d[rank - 1].is = is;
    d[rank - 1].os = os;
    for (i = rank - 1; i > 0; --i) {
        d[i - 1].n = n[i - 1];
        d[i - 1].is = d[i].is * niphys[i];
        d[i - 1].os = d[i].os * nophys[i];
    }
}

static int many_ok(int rank, const int *n, int howmany) {
    int i;
    if (howmany < 0 || rank < 0 || rank > FA_MAXRANK) return 0;   /* fftw_many_kosherp A.c:863-877 */
    for (i = 0; i < rank; ++i) if (n[i] <= 0) return 0;
    return 1;
}

/* default embedding of a real-data transform (reference fftw_rdft2_pad,
   fftw/fftw_api.c:774-788): the complex side always has n/2+1 along the last
   dim; the real side is padded to 2(n/2+1) only when in place */
static const int *rdft2_pad(int rank, const int *n, const int *nembed, int inplace, int cmplx,
                            int *store) {
    if (!nembed && rank > 0) {
        if (inplace || cmplx) {
            memcpy(store, n, sizeof(int) * (size_t)rank);
            store[rank - 1] = (n[rank - 1] / 2 + 1) * (cmplx ? 1 : 2);
            return store;
        }
        return n;
    }
    return nembed;
}

/* ------------------------------------------------- complex DFT planners */

fftw_plan fftw_plan_many_dft(int rank, const int *n, int howmany,
                             fftw_complex *in, const int *inembed, int istride, int idist,
                             fftw_complex *out, const int *onembed, int ostride, int odist,
                             int sign, unsigned flags) {
    fftw_iodim64 d[FA_MAXRANK], h;
    if (!many_ok(rank, n, howmany)) return NULL;
    rowmajor(rank, n, inembed ? inembed : n, onembed ? onembed : n, istride, ostride, d);
    h.n = howmany; h.is = idist; h.os = odist;
    return mk_guru(FA_C2C, rank, d, 1, &h, (double *)in, (double *)in + 1,
                   (double *)out, (double *)out + 1, 2, 2, sign, flags);
}

fftw_plan fftw_plan_dft(int rank, const int *n, fftw_complex *in, fftw_complex *out,
                        int sign, unsigned flags) {
    return fftw_plan_many_dft(rank, n, 1, in, 0, 1, 1, out, 0, 1, 1, sign, flags);
}

fftw_plan fftw_plan_dft_1d(int n, fftw_complex *in, fftw_complex *out, int sign, unsigned flags) {
    return fftw_plan_dft(1, &n, in, out, sign, flags);
}

fftw_plan fftw_plan_dft_2d(int n0, int n1, fftw_complex *in, fftw_complex *out,
                           int sign, unsigned flags) {
    int n[2];
    n[0] = n0; n[1] = n1;
    return fftw_plan_dft(2, n, in, out, sign, flags);
}

fftw_plan fftw_plan_dft_3d(int n0, int n1, int n2, fftw_complex *in, fftw_complex *out,
                           int sign, unsigned flags) {
    int n[3];
    n[0] = n0; n[1] = n1; n[2] = n2;
    return fftw_plan_dft(3, n, in, out, sign, flags);
}

fftw_plan fftw_plan_guru64_dft(int rank, const fftw_iodim64 *dims, int howmany_rank,
                               const fftw_iodim64 *howmany_dims, fftw_complex *in,
                               fftw_complex *out, int sign, unsigned flags) {
    return mk_guru(FA_C2C, rank, dims, howmany_rank, howmany_dims, (double *)in, (double *)in + 1,
                   (double *)out, (double *)out + 1, 2, 2, sign, flags);
}

fftw_plan fftw_plan_guru_dft(int rank, const fftw_iodim *dims, int howmany_rank,
                             const fftw_iodim *howmany_dims, fftw_complex *in, fftw_complex *out,
                             int sign, unsigned flags) {
    fftw_iodim64 d[FA_MAXRANK], h[FA_MAXRANK];
    if (rank < 0 || rank > FA_MAXRANK || howmany_rank < 0 || howmany_rank > FA_MAXRANK) return NULL;
    to64(rank, dims, d);
    to64(howmany_rank, howmany_dims, h);
    return fftw_plan_guru64_dft(rank, d, howmany_rank, h, in, out, sign, flags);
}

/* split arrays: strides count doubles; the transform is always forward
   (reference: "sign" of split plans is FFT_SIGN, fftw/fftw_api.c:1290-1330) */
fftw_plan fftw_plan_guru64_split_dft(int rank, const fftw_iodim64 *dims, int howmany_rank,
                                     const fftw_iodim64 *howmany_dims, double *ri, double *ii,
                                     double *ro, double *io, unsigned flags) {
    return mk_guru(FA_C2C, rank, dims, howmany_rank, howmany_dims, ri, ii, ro, io, 1, 1,
                   FFTW_FORWARD, flags);
}

fftw_plan fftw_plan_guru_split_dft(int rank, const fftw_iodim *dims, int howmany_rank,
                                   const fftw_iodim *howmany_dims, double *ri, double *ii,
                                   double *ro, double *io, unsigned flags) {
    fftw_iodim64 d[FA_MAXRANK], h[FA_MAXRANK];
    if (rank < 0 || rank > FA_MAXRANK || howmany_rank < 0 || howmany_rank > FA_MAXRANK) return NULL;
    to64(rank, dims, d);
    to64(howmany_rank, howmany_dims, h);
    return fftw_plan_guru64_split_dft(rank, d, howmany_rank, h, ri, ii, ro, io, flags);
}

/* ------------------------------------------------- real-data planners */

fftw_plan fftw_plan_many_dft_r2c(int rank, const int *n, int howmany,
                                 double *in, const int *inembed, int istride, int idist,
                                 fftw_complex *out, const int *onembed, int ostride, int odist,
                                 unsigned flags) {
    fftw_iodim64 d[FA_MAXRANK], h;
    int si[FA_MAXRANK], so[FA_MAXRANK], inplace;
    const int *ni, *no;
    if (!many_ok(rank, n, howmany) || rank < 1) return NULL;
    inplace = ((void *)in == (void *)out);
    ni = rdft2_pad(rank, n, inembed, inplace, 0, si);
    no = rdft2_pad(rank, n, onembed, inplace, 1, so);
    rowmajor(rank, n, ni, no, istride, ostride, d);
    h.n = howmany; h.is = idist; h.os = odist;
    return mk_guru(FA_R2C, rank, d, 1, &h, in, in, (double *)out, (double *)out + 1, 1, 2,
                   FFTW_FORWARD, flags);
}

fftw_plan fftw_plan_dft_r2c(int rank, const int *n, double *in, fftw_complex *out, unsigned flags) {
    return fftw_plan_many_dft_r2c(rank, n, 1, in, 0, 1, 1, out, 0, 1, 1, flags);
}
fftw_plan fftw_plan_dft_r2c_1d(int n, double *in, fftw_complex *out, unsigned flags) {
    return fftw_plan_dft_r2c(1, &n, in, out, flags);
}
fftw_plan fftw_plan_dft_r2c_2d(int n0, int n1, double *in, fftw_complex *out, unsigned flags) {
    int n[2];
    n[0] = n0; n[1] = n1;
    return fftw_plan_dft_r2c(2, n, in, out, flags);
}
fftw_plan fftw_plan_dft_r2c_3d(int n0, int n1, int n2, double *in, fftw_complex *out, unsigned flags) {
    int n[3];
    n[0] = n0; n[1] = n1; n[2] = n2;
    return fftw_plan_dft_r2c(3, n, in, out, flags);
}

fftw_plan fftw_plan_many_dft_c2r(int rank, const int *n, int howmany,
                                 fftw_complex *in, const int *inembed, int istride, int idist,
                                 double *out, const int *onembed, int ostride, int odist,
                                 unsigned flags) {
    fftw_iodim64 d[FA_MAXRANK], h;
    int si[FA_MAXRANK], so[FA_MAXRANK], inplace;
    const int *ni, *no;
    if (!many_ok(rank, n, howmany) || rank < 1) return NULL;
    inplace = ((void *)in == (void *)out);
    ni = rdft2_pad(rank, n, inembed, inplace, 1, si);
    no = rdft2_pad(rank, n, onembed, inplace, 0, so);
    rowmajor(rank, n, ni, no, istride, ostride, d);
    h.n = howmany; h.is = idist; h.os = odist;
    return mk_guru(FA_C2R, rank, d, 1, &h, (double *)in, (double *)in + 1, out, out, 2, 1,
                   FFTW_BACKWARD, flags);
}

fftw_plan fftw_plan_dft_c2r(int rank, const int *n, fftw_complex *in, double *out, unsigned flags) {
    return fftw_plan_many_dft_c2r(rank, n, 1, in, 0, 1, 1, out, 0, 1, 1, flags);
}
fftw_plan fftw_plan_dft_c2r_1d(int n, fftw_complex *in, double *out, unsigned flags) {
    return fftw_plan_dft_c2r(1, &n, in, out, flags);
}
fftw_plan fftw_plan_dft_c2r_2d(int n0, int n1, fftw_complex *in, double *out, unsigned flags) {
    int n[2];
    n[0] = n0; n[1] = n1;
    return fftw_plan_dft_c2r(2, n, in, out, flags);
}
fftw_plan fftw_plan_dft_c2r_3d(int n0, int n1, int n2, fftw_complex *in, double *out, unsigned flags) {
    int n[3];
    n[0] = n0; n[1] = n1; n[2] = n2;
    return fftw_plan_dft_c2r(3, n, in, out, flags);
}

fftw_plan fftw_plan_guru64_dft_r2c(int rank, const fftw_iodim64 *dims, int howmany_rank,
                                   const fftw_iodim64 *howmany_dims, double *in, fftw_complex *out,
                                   unsigned flags) {
    return mk_guru(FA_R2C, rank, dims, howmany_rank, howmany_dims, in, in, (double *)out,
                   (double *)out + 1, 1, 2, FFTW_FORWARD, flags);
}
fftw_plan fftw_plan_guru64_dft_c2r(int rank, const fftw_iodim64 *dims, int howmany_rank,
                                   const fftw_iodim64 *howmany_dims, fftw_complex *in, double *out,
                                   unsigned flags) {
    return mk_guru(FA_C2R, rank, dims, howmany_rank, howmany_dims, (double *)in, (double *)in + 1,
                   out, out, 2, 1, FFTW_BACKWARD, flags);
}
fftw_plan fftw_plan_guru64_split_dft_r2c(int rank, const fftw_iodim64 *dims, int howmany_rank,
                                         const fftw_iodim64 *howmany_dims, double *in, double *ro,
                                         double *io, unsigned flags) {
    return mk_guru(FA_R2C, rank, dims, howmany_rank, howmany_dims, in, in, ro, io, 1, 1,
                   FFTW_FORWARD, flags);
}
fftw_plan fftw_plan_guru64_split_dft_c2r(int rank, const fftw_iodim64 *dims, int howmany_rank,
                                         const fftw_iodim64 *howmany_dims, double *ri, double *ii,
                                         double *out, unsigned flags) {
    return mk_guru(FA_C2R, rank, dims, howmany_rank, howmany_dims, ri, ii, out, out, 1, 1,
                   FFTW_BACKWARD, flags);
}

#define GURU32(name64, ...)                                                                   \
    fftw_iodim64 d[FA_MAXRANK], h[FA_MAXRANK];                                                \
    if (rank < 0 || rank > FA_MAXRANK || howmany_rank < 0 || howmany_rank > FA_MAXRANK)       \
        return NULL;                                                                          \
    to64(rank, dims, d);                                                                      \
    to64(howmany_rank, howmany_dims, h);                                                      \
    return name64(rank, d, howmany_rank, h, __VA_ARGS__)

fftw_plan fftw_plan_guru_dft_r2c(int rank, const fftw_iodim *dims, int howmany_rank,
                                 const fftw_iodim *howmany_dims, double *in, fftw_complex *out,
                                 unsigned flags) {
    GURU32(fftw_plan_guru64_dft_r2c, in, out, flags);
}
fftw_plan fftw_plan_guru_dft_c2r(int rank, const fftw_iodim *dims, int howmany_rank,
                                 const fftw_iodim *howmany_dims, fftw_complex *in, double *out,
                                 unsigned flags) {
    GURU32(fftw_plan_guru64_dft_c2r, in, out, flags);
}
fftw_plan fftw_plan_guru_split_dft_r2c(int rank, const fftw_iodim *dims, int howmany_rank,
                                       const fftw_iodim *howmany_dims, double *in, double *ro,
                                       double *io, unsigned flags) {
    GURU32(fftw_plan_guru64_split_dft_r2c, in, ro, io, flags);
}
fftw_plan fftw_plan_guru_split_dft_c2r(int rank, const fftw_iodim *dims, int howmany_rank,
                                       const fftw_iodim *howmany_dims, double *ri, double *ii,
                                       double *out, unsigned flags) {
    GURU32(fftw_plan_guru64_split_dft_c2r, ri, ii, out, flags);
}

/* ---- r2r (reference fftw/fftw_api.c:738-840, 1235-1257, 1409-1430): every
   dim carries its own kind; strides are in doubles on both sides */
static plan *mk_r2r(int rank, const fftw_iodim64 *dims, int hrank, const fftw_iodim64 *hdims,
                    double *in, double *out, const fftw_r2r_kind *kind, unsigned flags) {
    plan *p;
    int i, k = 0, same = 1;
    if (!dims_ok(rank, dims) || !hdims_ok(hrank, hdims)) return NULL;
    if (rank > 0 && !kind) return NULL;
    for (i = 0; i < rank; ++i) {
        if ((int)kind[i] < FFTW_R2HC || (int)kind[i] > FFTW_RODFT11) return NULL;
        /* REDFT00 of one point has logical size 0: no solver applies in the reference either
           (redft00e applicable: n > 1, fftw/fftw_api.c:11842-11850) */
        if (kind[i] == FFTW_REDFT00 && dims[i].n < 2) return NULL;
    }
    p = fa_plan_new();
    if (!p) return NULL;
    p->type = FA_R2R;
    p->sign = FFTW_FORWARD;
    p->flags = flags;
    p->rank = rank;
    for (i = 0; i < rank; ++i) {
        p->dims[i].n = dims[i].n;
        p->dims[i].is = dims[i].is;
        p->dims[i].os = dims[i].os;
        p->kinds[i] = (int)kind[i];
        if (dims[i].is != dims[i].os) same = 0;
    }
    for (i = 0; i < hrank; ++i) {
        if (hdims[i].n == 1) continue;
        p->hdims[k].n = hdims[i].n;
        p->hdims[k].is = hdims[i].is;
        p->hdims[k].os = hdims[i].os;
        if (hdims[i].is != hdims[i].os) same = 0;
        ++k;
    }
    p->hrank = k;
    pick_batch(p);
    p->in_im = p->out_im = 0;
    /* in place with different layouts: one chunk, so that the first axis has read
       everything into scratch before anything is written back */
    if (in == out && !same) p->single_chunk = 1;
    return finish(p, in, in, out, out);
}

fftw_plan fftw_plan_guru64_r2r(int rank, const fftw_iodim64 *dims, int howmany_rank,
                               const fftw_iodim64 *howmany_dims, double *in, double *out,
                               const fftw_r2r_kind *kind, unsigned flags) {
    return mk_r2r(rank, dims, howmany_rank, howmany_dims, in, out, kind, flags);
}
fftw_plan fftw_plan_guru_r2r(int rank, const fftw_iodim *dims, int howmany_rank,
                             const fftw_iodim *howmany_dims, double *in, double *out,
                             const fftw_r2r_kind *kind, unsigned flags) {
    fftw_iodim64 d[FA_MAXRANK], h[FA_MAXRANK];
    if (rank < 0 || rank > FA_MAXRANK || howmany_rank < 0 || howmany_rank > FA_MAXRANK) return NULL;
    to64(rank, dims, d);
    to64(howmany_rank, howmany_dims, h);
    return mk_r2r(rank, d, howmany_rank, h, in, out, kind, flags);
}
fftw_plan fftw_plan_many_r2r(int rank, const int *n, int howmany, double *in, const int *inembed,
                             int istride, int idist, double *out, const int *onembed, int ostride,
                             int odist, const fftw_r2r_kind *kind, unsigned flags) {
    fftw_iodim64 d[FA_MAXRANK], h;
    if (!many_ok(rank, n, howmany)) return NULL;
    rowmajor(rank, n, inembed ? inembed : n, onembed ? onembed : n, istride, ostride, d);
    h.n = howmany; h.is = idist; h.os = odist;
    return mk_r2r(rank, d, 1, &h, in, out, kind, flags);
}
fftw_plan fftw_plan_r2r(int rank, const int *n, double *in, double *out,
                        const fftw_r2r_kind *kind, unsigned flags) {
    return fftw_plan_many_r2r(rank, n, 1, in, NULL, 1, 1, out, NULL, 1, 1, kind, flags);
}
fftw_plan fftw_plan_r2r_1d(int n, double *in, double *out, fftw_r2r_kind kind, unsigned flags) {
    return fftw_plan_r2r(1, &n, in, out, &kind, flags);
}
fftw_plan fftw_plan_r2r_2d(int n0, int n1, double *in, double *out, fftw_r2r_kind k0,
                           fftw_r2r_kind k1, unsigned flags) {
    int n[2];
    fftw_r2r_kind k[2];
    n[0] = n0; n[1] = n1;
    k[0] = k0; k[1] = k1;
    return fftw_plan_r2r(2, n, in, out, k, flags);
}
fftw_plan fftw_plan_r2r_3d(int n0, int n1, int n2, double *in, double *out, fftw_r2r_kind k0,
                           fftw_r2r_kind k1, fftw_r2r_kind k2, unsigned flags) {
    int n[3];
    fftw_r2r_kind k[3];
    n[0] = n0; n[1] = n1; n[2] = n2;
    k[0] = k0; k[1] = k1; k[2] = k2;
    return fftw_plan_r2r(3, n, in, out, k, flags);
}
void fftw_execute_r2r(const fftw_plan p, double *in, double *out) {
    fa_run(p, in, in, out, out);
}

/* ------------------------------------------------------------ execution */

/* reference fftw_execute, fftw/fftw_api.c:428-431 */
void fftw_execute(const fftw_plan p) {
    fa_run(p, p->ri, p->ii, p->ro, p->io);
}

/* new-array execution (reference fftw/fftw_api.c:434-440).  The arrays must
   have the layout and in-placeness the plan was created with. */
void fftw_execute_dft(const fftw_plan p, fftw_complex *in, fftw_complex *out) {
    fa_run(p, (double *)in, (double *)in + 1, (double *)out, (double *)out + 1);
}
void fftw_execute_split_dft(const fftw_plan p, double *ri, double *ii, double *ro, double *io) {
    fa_run(p, ri, ii, ro, io);
}
void fftw_execute_dft_r2c(const fftw_plan p, double *in, fftw_complex *out) {
    fa_run(p, in, in, (double *)out, (double *)out + 1);
}
void fftw_execute_dft_c2r(const fftw_plan p, fftw_complex *in, double *out) {
    fa_run(p, (double *)in, (double *)in + 1, out, out);
}
void fftw_execute_split_dft_r2c(const fftw_plan p, double *in, double *ro, double *io) {
    fa_run(p, in, in, ro, io);
}
void fftw_execute_split_dft_c2r(const fftw_plan p, double *ri, double *ii, double *out) {
    fa_run(p, ri, ii, out, out);
}

/* NULL-safe like the reference (fftw/fftw_api.c:409-410) */
void fftw_destroy_plan(fftw_plan p) { fa_plan_free(p); }

/* The only global planner state is the wisdom list (below); plans own their tables. */
void fftw_set_timelimit(double t) { (void)t; }
void fftw_plan_with_nthreads(int nthreads) { (void)nthreads; }
int  fftw_init_threads(void) { return 1; }
void fftw_cleanup_threads(void) {}
void fftw_make_planner_thread_safe(void) {}

/* ---- wisdom: text records "(key) chunk pipeline lmax bits ms", one per problem
   (bits: 1 = small_tiles, 2 = long_first, 4 | 8 = chunk lanes - 1, 16 = real_dec) */
void fftw_forget_wisdom(void) {
    pthread_mutex_lock(&g_planner_lock);
    while (g_wisdom) { wis_entry *n = g_wisdom->next; free(g_wisdom); g_wisdom = n; }
    pthread_mutex_unlock(&g_planner_lock);
}
void fftw_cleanup(void) { fftw_forget_wisdom(); }   /* plans stay valid, like the reference (A.c:411-418) */

char *fftw_export_wisdom_to_string(void) {
    size_t cap = 64, len = 0;
    wis_entry *w;
    char *s;
    pthread_mutex_lock(&g_planner_lock);
    for (w = g_wisdom; w; w = w->next) cap += strlen(w->key) + 96;
    s = (char *)malloc(cap);
    if (!s) { pthread_mutex_unlock(&g_planner_lock); return NULL; }
    len += (size_t)snprintf(s + len, cap - len, "(fftw3_amd_wisdom-1\n");
    for (w = g_wisdom; w; w = w->next)
        len += (size_t)snprintf(s + len, cap - len, "  (%s) %zu %d %d %d %.6f\n", w->key, w->cfg.chunk_bytes,
                                w->cfg.pipeline, w->cfg.lmax_multi, (w->cfg.small_tiles ? 1 : 0) | (w->cfg.long_first ? 2 : 0) | (((w->cfg.lanes > 1 ? w->cfg.lanes - 1 : 0) & 3) << 2) | (w->cfg.real_dec ? 16 : 0), w->ms);
    snprintf(s + len, cap - len, ")\n");
    pthread_mutex_unlock(&g_planner_lock);
    return s;
}
void fftw_export_wisdom_to_file(FILE *f) {
    char *s = fftw_export_wisdom_to_string();
    if (s) { fputs(s, f); free(s); }
}
int fftw_export_wisdom_to_filename(const char *filename) {
    FILE *f = fopen(filename, "w");
    if (!f) return 0;
    fftw_export_wisdom_to_file(f);
    return fclose(f) == 0;
}
void fftw_export_wisdom(fftw_write_char_func wr, void *data) {
    char *s = fftw_export_wisdom_to_string(), *c;
    if (!s) return;
    for (c = s; *c; ++c) wr(*c, data);
    free(s);
}

/* transactional like the reference (old wisdom survives a parse error, A.c:15577-15581) */
int fftw_import_wisdom_from_string(const char *input) {
    const char *c;
    wis_entry *staged = NULL, *w;
    int ok = 1;
    if (!input || strncmp(input, "(fftw3_amd_wisdom-1", 19)) return 0;
    c = input + 19;
    for (;;) {
        char key[FA_WIS_KEY_CAP];
        size_t chunk;
        int pipe, lmax, small, n = 0;
        double ms;
        const char *o, *e;
        while (*c == ' ' || *c == '\n' || *c == '\t' || *c == '\r') ++c;
        if (*c == ')') break;                       /* end of list */
        if (*c != '(') { ok = 0; break; }
        o = c + 1;
        e = strchr(o, ')');
        if (!e || (size_t)(e - o) >= sizeof(key)) { ok = 0; break; }
        memcpy(key, o, (size_t)(e - o));
        key[e - o] = 0;
        if (sscanf(e + 1, " %zu %d %d %d %lf%n", &chunk, &pipe, &lmax, &small, &ms, &n) != 5) { ok = 0; break; }
        if (chunk < ((size_t)1 << 16) || lmax < 16 || lmax > 1024) { ok = 0; break; }
        w = (wis_entry *)calloc(1, sizeof(*w));
        if (!w) { ok = 0; break; }
        snprintf(w->key, sizeof(w->key), "%s", key);
        w->cfg.chunk_bytes = chunk; w->cfg.pipeline = pipe != 0; w->cfg.lmax_multi = lmax; w->cfg.small_tiles = (small & 1) != 0; w->cfg.long_first = (small & 2) != 0; w->cfg.lanes = 1 + ((small >> 2) & 3); w->cfg.real_dec = (small & 16) != 0;
        w->ms = ms;
        w->next = staged;
        staged = w;
        c = e + 1 + n;
    }
    pthread_mutex_lock(&g_planner_lock);
    while (staged) {
        w = staged;
        staged = staged->next;
        if (ok) wis_put(w->key, w->cfg, w->ms);
        free(w);
    }
    pthread_mutex_unlock(&g_planner_lock);
    return ok;
}
int fftw_import_wisdom_from_file(FILE *f) {
    size_t cap = 4096, len = 0, n;
    char *buf = (char *)malloc(cap);
    int ok;
    if (!buf) return 0;
    while ((n = fread(buf + len, 1, cap - len - 1, f)) > 0) {
        len += n;
        if (len + 1 >= cap) { cap *= 2; buf = (char *)realloc(buf, cap); if (!buf) return 0; }
    }
    buf[len] = 0;
    ok = fftw_import_wisdom_from_string(buf);
    free(buf);
    return ok;
}
int fftw_import_wisdom_from_filename(const char *filename) {
    FILE *f = fopen(filename, "r");
    int ok;
    if (!f) return 0;
    ok = fftw_import_wisdom_from_file(f);
    fclose(f);
    return ok;
}
int fftw_import_system_wisdom(void) { return fftw_import_wisdom_from_filename("/etc/fftw/wisdom_amd"); }
int fftw_import_wisdom(fftw_read_char_func rd, void *data) {
    size_t cap = 4096, len = 0;
    char *buf = (char *)malloc(cap);
    int ch, ok;
    if (!buf) return 0;
    while ((ch = rd(data)) != EOF && ch > 0) {
        buf[len++] = (char)ch;
        if (len + 1 >= cap) { cap *= 2; buf = (char *)realloc(buf, cap); if (!buf) return 0; }
    }
    buf[len] = 0;
    ok = fftw_import_wisdom_from_string(buf);
    free(buf);
    return ok;
}

/* ---- introspection */
char *fftw_sprint_plan(const fftw_plan p) { return p ? fa_sprint(p) : NULL; }
void fftw_fprint_plan(const fftw_plan p, FILE *f) {
    char *s = fftw_sprint_plan(p);
    if (s) { fputs(s, f); free(s); }
}
void fftw_print_plan(const fftw_plan p) { fftw_fprint_plan(p, stdout); }

void fftw_flops(const fftw_plan p, double *add, double *mul, double *fmas) {
    double total = p->est_flops * (p->chunk ? (double)p->batch / (double)p->chunk : 0.0);
    *add = 0.6 * total;
    *mul = 0.2 * total;
    *fmas = 0.1 * total;
}
double fftw_estimate_cost(const fftw_plan p) {
    double a, m, f;
    fftw_flops(p, &a, &m, &f);
    return a + m + 2 * f;
}
double fftw_cost(const fftw_plan p) { return fftw_estimate_cost(p); }

/* ---- memory: pinned host memory when a device is present so that the
   staged path runs at PCIe DMA speed; plain aligned memory otherwise */
void *fftw_malloc(size_t n) {
    void *p = fa_hip_host_malloc(n);
    if (p) return p;
    if (posix_memalign(&p, 64, n ? n : 64)) return NULL;
    return p;
}
double *fftw_alloc_real(size_t n) { return (double *)fftw_malloc(n * sizeof(double)); }
fftw_complex *fftw_alloc_complex(size_t n) { return (fftw_complex *)fftw_malloc(n * sizeof(fftw_complex)); }
void fftw_free(void *p) {
    if (!p) return;
    if (!fa_hip_host_free(p)) free(p);
}
int fftw_alignment_of(double *p) { return (int)(((size_t)p) % 16); }

/* ------------------------------------------------- fftw_amd_ extensions */

int fftw_amd_device_count(void) { return fa_hip_device_count(); }
void *fftw_amd_malloc_device(size_t nbytes) {
    if (fa_hip_device_count() <= 0) return NULL;
    return fa_hip_malloc(nbytes);
}
void fftw_amd_free_device(void *p) { fa_hip_free(p); }
/* device of the calling host thread (hipSetDevice / hipGetDevice): what fftw_amd_malloc_device allocates on and
   what a plan created afterwards puts its tables and scratch on -- at plan creation (finish_locked ->
   fa_device_init), so a plan belongs to the device that was current when its planner function ran */
int fftw_amd_set_device(int device) {
    if (device < 0 || device >= fa_hip_device_count()) return -1;
    fa_hip_set_device(device);
    return 0;
}
int fftw_amd_get_device(void) { return fa_hip_device_count() > 0 ? fa_hip_get_device() : -1; }
/* blocking copies, so that a plain C caller of the device path needs no HIP headers */
void fftw_amd_memcpy_to_device(void *dst_device, const void *src_host, size_t nbytes) {
    if (!nbytes || fa_hip_device_count() <= 0) return;
    fa_hip_memcpy_h2d(dst_device, src_host, nbytes, NULL);
    fa_hip_stream_sync(NULL);
}
void fftw_amd_memcpy_to_host(void *dst_host, const void *src_device, size_t nbytes) {
    if (!nbytes || fa_hip_device_count() <= 0) return;
    fa_hip_memcpy_d2h(dst_host, src_device, nbytes, NULL);
    fa_hip_stream_sync(NULL);
}
void fftw_amd_plan_set_stream(fftw_plan p, void *s) { if (p) p->stream = s; }
void fftw_amd_plan_sync(fftw_plan p) { if (p && fa_hip_device_count() > 0) fa_hip_stream_sync(p->stream); }

size_t fftw_amd_plan_workspace_bytes(const fftw_plan p) {
    size_t b = 0;
    int i;
    for (i = 0; i < p->ntabs; ++i)
        b += (size_t)p->tabs[i].len * (p->tabs[i].kind == FA_TAB_PERM ? sizeof(i64) : 16);
    for (i = 2; i < p->nbufs; ++i) b += (size_t)p->buf_reals[i] * sizeof(double);
    return b;
}

/* device that holds the plan's tables and scratch: -1 before the device is set up or when the plan owns none;
   -2 if its allocations are NOT all on one device (a bug) */
int fftw_amd_plan_workspace_device(const fftw_plan p) {
    int i, dev = -1;
    if (!p || fa_hip_device_count() <= 0) return -1;
    for (i = 0; i < p->ntabs + p->nbufs; ++i) {
        const void *q = i < p->ntabs ? p->tabs[i].dev : (i - p->ntabs >= 2 ? (const void *)p->dbuf[i - p->ntabs] : NULL);
        int d;
        if (!q) continue;
        d = fa_hip_ptr_device(q);
        if (d < 0) continue;
        if (dev >= 0 && d != dev) return -2;
        dev = d;
    }
    return dev;
}

int fftw_amd_plan_num_steps(const fftw_plan p) { return p ? p->nsteps : 0; }
int fftw_amd_plan_get_step(const fftw_plan p, int i, fftw_amd_step_desc *out) {
    if (!p || i < 0 || i >= p->nsteps) return -1;
    *out = p->steps[i];
    return 0;
}
long long fftw_amd_plan_chunk(const fftw_plan p) { return p->chunk; }
int fftw_amd_plan_paired(const fftw_plan p) { return p ? p->pair : 0; }
int fftw_amd_plan_lanes(const fftw_plan p) { return (p && p->lanes > 1) ? p->lanes : 1; }
long long fftw_amd_plan_batch(const fftw_plan p) { return p->batch; }

/* returns the length in doubles (int64 tables are reported as doubles holding
   the integer values); negative kind-coded length -(src+1) - 1000000 is never
   used: a table the device has yet to compute reports 0 and its source id in
   dst[0] when cap >= 1 */
long long fftw_amd_plan_table(const fftw_plan p, int id, double *dst, long long cap) {
    const fa_table *t;
    long long i, n;
    if (!p || id < 0 || id >= p->ntabs) return -1;
    t = &p->tabs[id];
    if (t->kind == FA_TAB_PERM) {
        n = t->len;
        for (i = 0; i < n && i < cap; ++i) dst[i] = (double)((const i64 *)t->host)[i];
        return n;
    }
    if (!t->host) {
        if (cap >= 1) dst[0] = (double)t->src;
        return 0;
    }
    n = 2 * t->len;
    for (i = 0; i < n && i < cap; ++i) dst[i] = ((const double *)t->host)[i];
    return n;
}

void fftw_amd_cexp(long long m, long long n, double out[2]) { fa_cexp(m, n, out); }
long long fftw_amd_find_generator(long long p) { return fa_find_generator(p); }
long long fftw_amd_power_mod(long long b, long long e, long long p) { return fa_power_mod(b, e, p); }
int fftw_amd_factor_passes(long long n, int max_passes, long long *lens) {
    return fa_factor_passes(n, max_passes, FA_LMAX_SINGLE, 1024, lens);
}
