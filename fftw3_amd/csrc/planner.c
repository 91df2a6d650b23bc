/*
 * planner.c -- the static GPU planner: turns a DFT problem (dims + strides)
 * into a short list of kernel launches ("steps").
 *
 * It replaces the reference's search-based planner (ifftw_mkplan,
 * fftw/fftw_api.c:15300-15426) and restates the *structure* of the solvers it
 * would have picked:
 *   - Cooley-Tukey n = L1*L2*...  (ct_mkplan A.c:2183-2202)  -> fa_emit_ct
 *   - vector / rank loops (vrank_geq1 A.c:4627, rank_geq2 A.c:4435) -> loop
 *     dims carried by every step and executed by the kernel grid
 *   - Rader (A.c:4187-4261) and Bluestein (A.c:1642-1688) for sizes with a
 *     prime factor the LDS kernel has no stage for
 *   - rdft2 via half-length complex DFT + untangle (ct_hc2c A.c:5579-5590)
 * A CPU cost model is meaningless on the GPU, so there is no search: the
 * decomposition is a function of n and of which index is contiguous.
 */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fa_plan.h"
#include "fa_hip.h"

typedef struct fftw_plan_s plan;

/* scratch per chunk: what one chunk writes between two passes must still be in the 256 MiB
   Infinity Cache when the next pass reads it.  Round 3 (tests/micro/mall_probe.hip, profiles/r03_mall_probe.txt):
   the two-pass 2^20 plan as pure data movement takes 10.0 us per transform as serial launches over 256 MiB
   chunks, and 8.2 us when chunk c runs on stream c % 2 with 96 ... 128 MiB of scratch per chunk (two lanes,
   256 MiB of scratch in flight): the tail of one lane's launch is filled by the other lane's, and the
   reuse distance of a scratch line is a few transforms instead of a whole 1 GiB launch pair. */
#define FA_DEFAULT_CHUNK_BYTES ((size_t)128 << 20)
/* The only process-wide planner setting: the documented setter below (read and written atomically).
   Everything else a plan is built from lives in the plan's own fa_cfg; the FFTW_AMD_* environment
   overrides are read into that copy at plan creation, never into shared state, so threads that plan
   with different settings do not see each other's. */
static size_t g_chunk_bytes = FA_DEFAULT_CHUNK_BYTES;

void fftw_amd_set_chunk_bytes(size_t nbytes) {
    __atomic_store_n(&g_chunk_bytes, nbytes ? nbytes : FA_DEFAULT_CHUNK_BYTES, __ATOMIC_RELAXED);
}

static i64 iabs(i64 v) { return v < 0 ? -v : v; }

fa_cfg fa_default_cfg(void) {
    fa_cfg c;
    const char *e;
    c.chunk_bytes = __atomic_load_n(&g_chunk_bytes, __ATOMIC_RELAXED);
    c.pipeline = 0;       /* two-stream chunk pipeline: off under FFTW_ESTIMATE (each kernel then owns the machine and
                             its launch duration is its roofline), a FFTW_MEASURE candidate, FFTW_AMD_PIPELINE=1 */
    c.lmax_multi = 1024;
    c.small_tiles = 1;
    c.long_first = 0;
    c.tile_elems = 0;
    c.real_dec = 0;
    c.lanes = 2;
    e = getenv("FFTW_AMD_CHUNK_BYTES");
    if (e && atoll(e) > 0) c.chunk_bytes = (size_t)atoll(e);
    e = getenv("FFTW_AMD_SMALL_TILES");
    if (e) c.small_tiles = atoi(e);
    e = getenv("FFTW_AMD_TILE_ELEMS");
    if (e) c.tile_elems = atoll(e);
    e = getenv("FFTW_AMD_LONG_FIRST");
    if (e) c.long_first = atoi(e);
    e = getenv("FFTW_AMD_PIPELINE");
    if (e) c.pipeline = atoi(e);
    e = getenv("FFTW_AMD_LMAX_MULTI");
    if (e && atoll(e) >= 16) c.lmax_multi = (int)atoll(e);
    e = getenv("FFTW_AMD_REAL_DEC");
    if (e) c.real_dec = atoi(e) != 0;
    e = getenv("FFTW_AMD_LANES");
    if (e && atoi(e) >= 1 && atoi(e) <= FA_MAXLANES) c.lanes = atoi(e);
    return c;
}

/* ------------------------------------------------------------------ plan */

plan *fa_plan_new(void) {
    plan *p = (plan *)calloc(1, sizeof(plan));
    if (!p) return NULL;
    p->nbufs = 2;
    p->in_im = p->out_im = 1;
    p->cfg = fa_default_cfg();
    pthread_mutex_init(&p->lock, NULL);
    return p;
}

static void plan_reset_build(plan *p) {
    int i;
    free(p->steps);
    p->steps = NULL;
    p->nsteps = p->cap_steps = 0;
    for (i = 0; i < p->ntabs; ++i) free(p->tabs[i].host);
    memset(p->tabs, 0, sizeof(p->tabs));
    p->ntabs = 0;
    p->nbufs = 2;
    memset(p->buf_reals, 0, sizeof(p->buf_reals));
    memset(p->buf_busy, 0, sizeof(p->buf_busy));
    p->failed = 0;
    p->est_flops = 0;
}

void fa_plan_free(plan *p) {
    int i;
    if (!p) return;
    if (p->alt) { fa_plan_free(p->alt); p->alt = NULL; }
    {
        /* device resources may exist without dev_ready (fa_device_init gave up half way: out of device memory) */
        int any = p->dev_ready || p->stage_in || p->stage_out || p->hstream[0] || p->pstream[0];
        for (i = 0; i < p->ntabs && !any; ++i) any = p->tabs[i].dev != NULL;
        for (i = 2; i < p->nbufs && !any; ++i) any = p->dbuf[i] != NULL;
        if (!any) goto host_only;
    }
    {
        fa_hip_stream_sync(p->stream);
        for (i = 0; i < p->ntabs; ++i) fa_hip_free(p->tabs[i].dev);
        for (i = 2; i < p->nbufs; ++i) fa_hip_free(p->dbuf[i]);
        fa_hip_free(p->stage_in);
        fa_hip_free(p->stage_out);
        if (p->hstream[0]) fa_hip_stream_destroy(p->hstream[0]);
        if (p->hstream[1]) fa_hip_stream_destroy(p->hstream[1]);
        if (p->pstream[0]) {
            for (i = 0; i < 4; ++i) { if (p->ev_a[i]) fa_hip_event_destroy(p->ev_a[i]); if (p->ev_b[i]) fa_hip_event_destroy(p->ev_b[i]); }
            fa_hip_event_destroy(p->ev_begin);
            for (i = 0; i < FA_MAXLANES; ++i) {
                if (p->ev_end[i]) fa_hip_event_destroy(p->ev_end[i]);
                if (p->pstream[i]) fa_hip_stream_destroy(p->pstream[i]);
            }
        }
    }
host_only:
    for (i = 0; i < p->ntabs; ++i) free(p->tabs[i].host);
    free(p->steps);
    pthread_mutex_destroy(&p->lock);
    free(p);
}

/* ---------------------------------------------------------------- tables */

static int tab_find(plan *p, int kind, i64 n, i64 aux) {
    int i;
    for (i = 0; i < p->ntabs; ++i)
        if (p->tabs[i].kind == kind && p->tabs[i].n == n && p->tabs[i].aux == aux) return i;
    return -1;
}

static int tab_new(plan *p, int kind, i64 n, i64 aux, i64 len, size_t elem) {
    fa_table *t;
    if (p->ntabs >= FA_MAXTAB) { p->failed = 1; return 0; }
    t = &p->tabs[p->ntabs];
    t->kind = kind;
    t->n = n;
    t->aux = aux;
    t->len = len;
    t->src = -1;
    t->host = len ? calloc((size_t)len, elem) : NULL;
    t->dev = NULL;
    return p->ntabs++;
}

static int tab_stage(plan *p, i64 L) {
    int id = tab_find(p, FA_TAB_STAGE, L, 0);
    double *h;
    i64 m;
    if (id >= 0) return id;
    id = tab_new(p, FA_TAB_STAGE, L, 0, L, 2 * sizeof(double));
    h = (double *)p->tabs[id].host;
    for (m = 0; m < L; ++m) fa_cexp(m, L, h + 2 * m);
    return id;
}

/* w^m = lo[m & (2^s - 1)] * hi[m >> s], both factors exact table entries:
   the reference's sqrt(n) scheme for large radix steps
   (fftw/fftw_api.c:18989-19011, rotate A.c:18920-18941) */
static void tab_tw2(plan *p, i64 n, int *lo, int *hi, int *shift) {
    int s = 0, id;
    i64 m, nlo, nhi;
    double *h;
    while (((i64)1 << (2 * s)) < n) ++s;
    nlo = (i64)1 << s;
    nhi = (n + nlo - 1) / nlo;
    *shift = s;
    id = tab_find(p, FA_TAB_TW_LO, n, s);
    if (id < 0) {
        id = tab_new(p, FA_TAB_TW_LO, n, s, nlo, 2 * sizeof(double));
        h = (double *)p->tabs[id].host;
        for (m = 0; m < nlo; ++m) fa_cexp(m, n, h + 2 * m);
    }
    *lo = id;
    id = tab_find(p, FA_TAB_TW_HI, n, s);
    if (id < 0) {
        id = tab_new(p, FA_TAB_TW_HI, n, s, nhi, 2 * sizeof(double));
        h = (double *)p->tabs[id].host;
        for (m = 0; m < nhi; ++m) fa_cexp(m << s, n, h + 2 * m);
    }
    *hi = id;
}

/* Bluestein chirp w[k] = exp(+i pi k^2 / n), k^2 reduced mod 2n first
   (reference bluestein_sequence, fftw/fftw_api.c:1598-1611) */
static int tab_chirp(plan *p, i64 n, i64 nb) {
    int id = tab_find(p, FA_TAB_CHIRP, n, nb);
    double *h;
    i64 k;
    if (id >= 0) return id;
    id = tab_new(p, FA_TAB_CHIRP, n, nb, nb, 2 * sizeof(double));
    h = (double *)p->tabs[id].host;
    for (k = 0; k < n; ++k) fa_cexp(fa_mulmod(k, k, 2 * n), 2 * n, h + 2 * k);
    return id;
}

static int tab_dft_of(plan *p, int src, i64 n) {
    int id = tab_find(p, FA_TAB_DFT_OF, n, src);
    if (id >= 0) return id;
    id = tab_new(p, FA_TAB_DFT_OF, n, src, n, 2 * sizeof(double));
    free(p->tabs[id].host);
    p->tabs[id].host = NULL;
    p->tabs[id].src = src;
    return id;
}

/* DFT_nb( w[0], w[1..n-1], 0.., w[n-1..1] ) / nb  (A.c:1624-1639) */
static int tab_blue_kernel(plan *p, i64 n, i64 nb) {
    int seq = tab_find(p, FA_TAB_BLUE_SEQ, n, nb);
    if (seq < 0) {
        double *h, w[2];
        i64 k;
        seq = tab_new(p, FA_TAB_BLUE_SEQ, n, nb, nb, 2 * sizeof(double));
        h = (double *)p->tabs[seq].host;
        for (k = 0; k < n; ++k) {
            fa_cexp(fa_mulmod(k, k, 2 * n), 2 * n, w);
            h[2 * k] = w[0] / (double)nb;
            h[2 * k + 1] = w[1] / (double)nb;
            if (k) {
                h[2 * (nb - k)] = h[2 * k];
                h[2 * (nb - k) + 1] = h[2 * k + 1];
            }
        }
    }
    return tab_dft_of(p, seq, nb);
}

/* perm[k] = g^k mod p (inverse = 0) or g^-k mod p (inverse = 1), k in [0, p-1) */
static int tab_perm(plan *p, i64 pr, int inverse) {
    int id = tab_find(p, FA_TAB_PERM, pr, inverse);
    i64 *h, g, x, k;
    if (id >= 0) return id;
    id = tab_new(p, FA_TAB_PERM, pr, inverse, pr - 1, sizeof(i64));
    h = (i64 *)p->tabs[id].host;
    g = fa_find_generator(pr);
    if (inverse) g = fa_power_mod(g, pr - 2, pr);
    x = 1;
    for (k = 0; k < pr - 1; ++k) {
        h[k] = x;
        x = fa_mulmod(x, g, pr);
    }
    return id;
}

/* Omega = DFT_{p-1}( exp(-2 pi i g^-j / p) / (p-1) )   (rader_mkomega A.c:4139-4167) */
static int tab_rader_kernel(plan *p, i64 pr) {
    int seq = tab_find(p, FA_TAB_RADER_SEQ, pr, 0);
    if (seq < 0) {
        int pi = tab_perm(p, pr, 1);
        const i64 *perm = (const i64 *)p->tabs[pi].host;
        double *h, w[2];
        i64 j;
        seq = tab_new(p, FA_TAB_RADER_SEQ, pr, 0, pr - 1, 2 * sizeof(double));
        h = (double *)p->tabs[seq].host;
        for (j = 0; j < pr - 1; ++j) {
            fa_cexp(perm[j], pr, w);
            h[2 * j] = w[0] / (double)(pr - 1);
            h[2 * j + 1] = -w[1] / (double)(pr - 1);
        }
    }
    return tab_dft_of(p, seq, pr - 1);
}

/* --------------------------------------------------------------- buffers */

static int buf_acquire(plan *p, i64 reals) {
    int i;
    for (i = 2; i < p->nbufs; ++i)
        if (!p->buf_busy[i]) {
            p->buf_busy[i] = 1;
            if (p->buf_reals[i] < reals) p->buf_reals[i] = reals;
            return i;
        }
    if (p->nbufs >= FA_MAXBUF) { p->failed = 1; return 2; }
    i = p->nbufs++;
    p->buf_busy[i] = 1;
    p->buf_reals[i] = reals;
    return i;
}

static void buf_release(plan *p, int id) { p->buf_busy[id] = 0; }

/* ----------------------------------------------------------------- steps */

static fftw_amd_step_desc *new_step(plan *p, int kind) {
    fftw_amd_step_desc *s;
    if (p->nsteps == p->cap_steps) {
        p->cap_steps = p->cap_steps ? 2 * p->cap_steps : 16;
        p->steps = (fftw_amd_step_desc *)realloc(p->steps, sizeof(*s) * (size_t)p->cap_steps);
    }
    s = &p->steps[p->nsteps++];
    memset(s, 0, sizeof(*s));
    s->kind = kind;
    s->table = s->table2 = -1;
    s->tw_lo = s->tw_hi = -1;
    s->batch_dim = -1;
    s->aux_buf = -1;
    s->tile = 1;
    s->tile_lo_n = 1;
    s->variant = FFTW_AMD_K_GENERIC;
    return s;
}

typedef struct { i64 n, is, os, tw; int is_batch; } sdim;

/* put dims into the step; returns -1 if they do not fit */
static int step_set_dims(plan *p, fftw_amd_step_desc *s, const sdim *d, int nd, int tile_idx) {
    int i, k = 0;
    if (tile_idx >= 0) {
        s->dim_n[0] = d[tile_idx].n; s->dim_is[0] = d[tile_idx].is;
        s->dim_os[0] = d[tile_idx].os; s->dim_tw[0] = d[tile_idx].tw;
        if (d[tile_idx].is_batch) s->batch_dim = 0;
        k = 1;
    }
    for (i = 0; i < nd; ++i) {
        if (i == tile_idx) continue;
        if (d[i].n == 1 && !d[i].is_batch) continue;
        if (k >= FFTW_AMD_MAX_DIMS) { p->failed = 1; return -1; }
        s->dim_n[k] = d[i].n; s->dim_is[k] = d[i].is;
        s->dim_os[k] = d[i].os; s->dim_tw[k] = d[i].tw;
        if (d[i].is_batch) s->batch_dim = k;
        ++k;
    }
    s->ndims = k;
    return 0;
}


/* Order the loops of an element-wise step by their stride on the user side (by_dst: the
   destination, else the source), the batch loop last, and return how many of them are
   faster than the transform index of stride kstride (step.kpos).  Extent-1 loops drop out
   here so that the count matches what step_set_dims keeps. */
static int order_dims_kpos(sdim *d, int *nd_io, int by_dst, i64 kstride) {
    sdim t[FA_MAXLOOPS + 1], b;
    int nd = *nd_io, cnt = 0, i, j, have_b = 0, kpos = 0;
    for (i = 0; i < nd; ++i) {
        i64 key;
        if (d[i].is_batch) { b = d[i]; have_b = 1; continue; }
        if (d[i].n == 1) continue;
        key = iabs(by_dst ? d[i].os : d[i].is);
        for (j = cnt - 1; j >= 0 && iabs(by_dst ? t[j].os : t[j].is) > key; --j) t[j + 1] = t[j];
        t[j + 1] = d[i];
        ++cnt;
    }
    for (i = 0; i < cnt; ++i) if (iabs(by_dst ? t[i].os : t[i].is) < iabs(kstride)) kpos = i + 1;
    if (have_b) t[cnt++] = b;
    memcpy(d, t, sizeof(sdim) * (size_t)cnt);
    *nd_io = cnt;
    return kpos;
}

/* One batched length-L DFT pass.  dims: every index other than the transform
   index.  The tile dim is the one whose stride makes global access widest. */
static void emit_pass(plan *p, fa_loc src, fa_loc dst, i64 L, i64 is_l, i64 os_l,
                      const sdim *dims, int nd, i64 tw_n, int flags) {
    fftw_amd_step_desc *s = new_step(p, FFTW_AMD_STEP_PASS);
    int i, best = -1, pair = -1;
    i64 best_cost = 0, T;
    const i64 INF = (i64)1 << 60;
    sdim dummy, dims_local[FFTW_AMD_MAX_DIMS + FA_MAXLOOPS];
    /* the pair loop of interleaved vectors is reserved for the inner tile component */
    for (i = 0; i < nd; ++i)
        if (dims[i].n == 2 && !dims[i].is_batch && dims[i].tw == 0 &&
            (iabs(dims[i].is) == 2 || iabs(dims[i].os) == 2) &&
            (dims[i].is % 2) == 0 && (dims[i].os % 2) == 0) { pair = i; break; }
    if (L * 3 > FA_LDS_ELEMS || FA_TILE_ELEMS / L < 2) pair = -1;   /* a tile must hold both members */
    for (i = 0; i < nd; ++i) {
        i64 ci, co, cost;
        if (dims[i].n <= 1 || i == pair) continue;
        ci = dims[i].is ? iabs(dims[i].is) : INF;
        co = dims[i].os ? iabs(dims[i].os) : INF;
        if (L > 1 && iabs(is_l) < ci) ci = iabs(is_l);
        if (L > 1 && iabs(os_l) < co) co = iabs(os_l);
        cost = ci + co;
        if (best < 0 || cost < best_cost ||
            (cost == best_cost && dims[i].n > dims[best].n)) {
            best = i;
            best_cost = cost;
        }
    }
    /* a two-element loop whose stride is one complex on either side (the two
       interleaved vectors of a radix-4 real transform) becomes the inner
       component of the tile dim: its pairs then share 32-byte accesses */
    if (best < 0 && pair >= 0) { best = pair; pair = -1; }   /* nothing else to tile over */
    {
        sdim rest[FFTW_AMD_MAX_DIMS + FA_MAXLOOPS];
        int lo = pair, k2 = 0;
        if (lo >= 0) {
            int nb = best;
            s->tile_lo_n = 2;
            s->tile_lo_is = dims[lo].is;
            s->tile_lo_os = dims[lo].os;
            for (i = 0; i < nd; ++i) {
                if (i == lo) { if (i < best) --nb; continue; }
                rest[k2++] = dims[i];
            }
            memcpy((void *)dims_local, rest, sizeof(sdim) * (size_t)k2);
            dims = dims_local;
            nd = k2;
            best = nb;
        }
    }
    s->src_buf = src.buf; s->src_base = src.base; s->src_im = src.im;
    s->dst_buf = dst.buf; s->dst_base = dst.base; s->dst_im = dst.im;
    s->flags = flags;
    s->L = (int)L;
    s->is_l = is_l;
    s->os_l = os_l;
    s->nradices = fa_radices(L, s->radices);
    if (s->nradices < 0) { p->failed = 1; return; }
    if (L > 1) s->table = tab_stage(p, L);
    if (best < 0) {
        /* no loop with extent > 1: a dummy tile dim keeps the kernel uniform,
           any batch dim of extent 1 still rides along as a loop */
        sdim tmp[FFTW_AMD_MAX_DIMS + 1];
        memset(&dummy, 0, sizeof(dummy));
        dummy.n = 1;
        tmp[0] = dummy;
        for (i = 0; i < nd && i < FFTW_AMD_MAX_DIMS; ++i) tmp[i + 1] = dims[i];
        step_set_dims(p, s, tmp, nd + 1, 0);
    } else {
        step_set_dims(p, s, dims, nd, best);
    }
    T = FA_TILE_ELEMS / L;
    /* Small tiles for the LDS kernels: about 1024 elements per workgroup (32 KiB of
       ping-pong LDS, 4-5 workgroups per CU) hide the latencies of the stage loop far
       better than one 128 KiB workgroup -- measured 1.46 -> 2.30 TB/s for n = 1000,
       0.96 -> 1.35 TB/s for n = 5000.  Strided passes keep 8
       sequences (128-byte segments).  small_tiles = 0 (a FFTW_MEASURE candidate) keeps
       the 4096-element tiles.  The register kernels set their own tile below. */
    {
        i64 elems = p->cfg.tile_elems > 0 ? p->cfg.tile_elems : (p->cfg.small_tiles ? 1024 : FA_TILE_ELEMS);
        i64 cap = elems / L, floor_t = (iabs(is_l) <= 2 && iabs(os_l) <= 2) ? 1 : 8;
        if (cap < floor_t) cap = floor_t;
        if (T > cap) T = cap;
    }
    if (T < 1) T = 1;
    if (T > s->dim_n[0] * s->tile_lo_n) T = s->dim_n[0] * s->tile_lo_n;
    /* the LDS row is padded to an odd width (T | 1): both images must fit 160 KiB */
    while (T > 1 && L * (T | 1) > FA_LDS_ELEMS) --T;
    T -= T % s->tile_lo_n;
    if (T < 1) T = 1;
    s->tile = (int)T;
    if (src.im == 1 && dst.im == 1 && !(flags & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT)) &&
        !getenv("FFTW_AMD_NO_TUNED")) {
        if (L == 1024) {
            s->variant = FFTW_AMD_K_P1024;  /* register-resident radix-32x32, 8 sequences per tile */
            s->tile = 8;
        } else if (L <= 32 && fa_hip_r1_tile((int)L) > 0 && is_l == 2 && os_l == 2 && tw_n == 0 && s->tile_lo_n == 1 &&
                   s->dim_is[0] == 2 * L && s->dim_os[0] == 2 * L && s->dim_n[0] >= 256 && !getenv("FFTW_AMD_NO_R1")) {
            s->variant = FFTW_AMD_K_R1;     /* dense rows of a short length: one butterfly per row (pass1r.hpp) */
            s->tile = fa_hip_r1_tile((int)L);
        } else if (fa_hip_r3_tile((int)L) > 0 && is_l == 2 && os_l == 2 && tw_n == 0 && s->tile_lo_n == 1 &&
                   (L == 2048 || L == 4096 || L == 8192 || L == 16384 || s->dim_n[0] * 2 >= fa_hip_r3_tile((int)L))) {
            s->variant = FFTW_AMD_K_R3;     /* three-stage register kernel, whole contiguous rows */
            s->tile = fa_hip_r3_tile((int)L);
        } else if (fa_hip_rr_tile((int)L) > 0 && s->dim_n[0] * s->tile_lo_n * 4 >= fa_hip_rr_tile((int)L) &&
                   (s->tile_lo_n == 1 || (L & (L - 1)) == 0)) {
            s->variant = FFTW_AMD_K_RR;     /* two-stage register kernel */
            s->tile = fa_hip_rr_tile((int)L);
        } else if (fa_hip_r3t_tile((int)L) > 0 && s->tile_lo_n == 1 && s->dim_n[0] * 4 >= fa_hip_r3t_tile((int)L)) {
            s->variant = FFTW_AMD_K_R3;     /* three-stage register kernel, column / transposed form */
            s->tile = fa_hip_r3t_tile((int)L);
        }
    }
    s->tw_n = tw_n;
    if (tw_n) tab_tw2(p, tw_n, &s->tw_lo, &s->tw_hi, &s->tw_shift);
    {
        double work = (double)L;
        for (i = 0; i < s->ndims; ++i) work *= (double)s->dim_n[i];
        if (L > 1) p->est_flops += 5.0 * work * log2((double)L);
        if (tw_n) p->est_flops += 6.0 * work;
    }
}

static void emit_copy(plan *p, int kind, fa_loc src, fa_loc dst, i64 K, i64 Kvalid,
                      i64 is_k, i64 os_k, const sdim *dims, int nd, int flags,
                      int table, int table2) {
    fftw_amd_step_desc *s = new_step(p, kind);
    s->src_buf = src.buf; s->src_base = src.base; s->src_im = src.im;
    s->dst_buf = dst.buf; s->dst_base = dst.base; s->dst_im = dst.im;
    s->flags = flags;
    s->is_l = is_k;
    s->os_l = os_k;
    s->aux_n = K;
    s->aux_valid = Kvalid;
    s->table = table;
    s->table2 = table2;
    step_set_dims(p, s, dims, nd, -1);
}

/* ------------------------------------------------------------- one axis */

static void fa_emit_axis(plan *p, const fa_axis *ax);
static int has_register_kernel(i64 L);

static int loops_to_sdims(const fa_axis *ax, sdim *d, int use_dst_as_src) {
    int i;
    for (i = 0; i < ax->nloops; ++i) {
        d[i].n = ax->loops[i].n;
        d[i].is = use_dst_as_src ? ax->loops[i].os : ax->loops[i].is;
        d[i].os = ax->loops[i].os;
        d[i].tw = 0;
        d[i].is_batch = (i == ax->batch_loop);
    }
    return ax->nloops;
}

/* dense scratch layout for an axis: batch loop outermost, then the remaining
   indices in the order of their source strides, innermost stride 2 (one
   interleaved complex).  Returns total doubles; fills axis and loop strides. */
static i64 scratch_layout_u(const fa_axis *ax, i64 n_axis, i64 unit, i64 *ts_axis, i64 *ts_loop) {
    int order[FA_MAXLOOPS + 1], cnt = 0, i, j;
    i64 key[FA_MAXLOOPS + 1], stride = unit;
    /* index -1 denotes the axis itself; it comes first so that it wins ties (callers pass
       is = 1 to ask for the axis innermost, and a loop of real data may have stride 1 too) */
    order[cnt] = -1;
    key[cnt] = iabs(ax->is);
    ++cnt;
    for (i = 0; i < ax->nloops; ++i) {
        if (i == ax->batch_loop) continue;
        order[cnt] = i;
        key[cnt] = iabs(ax->loops[i].is);
        ++cnt;
    }
    /* insertion sort ascending by source stride: smallest stride innermost */
    for (i = 1; i < cnt; ++i) {
        int o = order[i];
        i64 k = key[i];
        for (j = i - 1; j >= 0 && key[j] > k; --j) { order[j + 1] = order[j]; key[j + 1] = key[j]; }
        order[j + 1] = o;
        key[j + 1] = k;
    }
    for (i = 0; i < cnt; ++i) {
        if (order[i] < 0) { *ts_axis = stride; stride *= n_axis; }
        else { ts_loop[order[i]] = stride; stride *= ax->loops[order[i]].n; }
    }
    if (ax->batch_loop >= 0) {
        ts_loop[ax->batch_loop] = stride;
        stride *= ax->loops[ax->batch_loop].n;
    }
    return stride;
}

static i64 scratch_layout(const fa_axis *ax, i64 n_axis, i64 *ts_axis, i64 *ts_loop) {
    return scratch_layout_u(ax, n_axis, 2, ts_axis, ts_loop);
}

/* Cooley-Tukey over k >= 2 passes through a scratch image:
   pass i < k : length-L_i DFTs down columns of the [L_i][M_i] view, times
               w_{L_i M_i}^(k_i j);   last pass: length-L_k DFTs along rows,
   written transposed so the output is in natural order. */
static void emit_ct(plan *p, const fa_axis *ax, const i64 *lens, int k) {
    i64 M[FA_MAXPASS + 1], Pf[FA_MAXPASS + 1];
    i64 ts, lts[FA_MAXLOOPS], total;
    sdim d[FFTW_AMD_MAX_DIMS + FA_MAXLOOPS];
    fa_loc tmp;
    int i, j, nd, tbuf;

    /* M[i] = product of the lengths after pass i, Pf[i] = product before it */
    M[k - 1] = 1;
    for (i = k - 2; i >= 0; --i) M[i] = M[i + 1] * lens[i + 1];
    Pf[0] = 1;
    for (i = 1; i < k; ++i) Pf[i] = Pf[i - 1] * lens[i - 1];

    total = scratch_layout(ax, ax->n, &ts, lts);
    tbuf = buf_acquire(p, total);
    tmp.buf = tbuf;
    tmp.base = 0;
    tmp.im = 1;

    for (i = 0; i < k; ++i) {
        int flags = 0;
        nd = 0;
        if (i == 0) {
            /* source -> scratch */
            d[nd].n = M[0]; d[nd].is = ax->is; d[nd].os = ts; d[nd].tw = 1; d[nd].is_batch = 0; ++nd;
            for (j = 0; j < ax->nloops; ++j) {
                d[nd].n = ax->loops[j].n; d[nd].is = ax->loops[j].is; d[nd].os = lts[j];
                d[nd].tw = 0; d[nd].is_batch = (j == ax->batch_loop); ++nd;
            }
            flags = ax->flags_in & (FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_REAL_IN);
            /* two passes: the twiddle w_n^(k_0 j) is applied by the last pass on
               its input (index j along the row, k_0 = which row) */
            emit_pass(p, ax->src, tmp, lens[0], M[0] * ax->is, M[0] * ts, d, nd,
                      k == 2 ? 0 : ax->n, flags);
        } else if (i < k - 1) {
            /* scratch -> scratch, in place */
            d[nd].n = M[i]; d[nd].is = ts; d[nd].os = ts; d[nd].tw = 1; d[nd].is_batch = 0; ++nd;
            for (j = 0; j < i; ++j) {
                d[nd].n = lens[j]; d[nd].is = M[j] * ts; d[nd].os = M[j] * ts;
                d[nd].tw = 0; d[nd].is_batch = 0; ++nd;
            }
            for (j = 0; j < ax->nloops; ++j) {
                d[nd].n = ax->loops[j].n; d[nd].is = lts[j]; d[nd].os = lts[j];
                d[nd].tw = 0; d[nd].is_batch = (j == ax->batch_loop); ++nd;
            }
            emit_pass(p, tmp, tmp, lens[i], M[i] * ts, M[i] * ts, d, nd, lens[i] * M[i], 0);
        } else {
            /* scratch -> destination, digits reversed into natural order */
            for (j = 0; j < k - 1; ++j) {
                d[nd].n = lens[j]; d[nd].is = M[j] * ts; d[nd].os = Pf[j] * ax->os;
                d[nd].tw = (k == 2) ? 1 : 0; d[nd].is_batch = 0; ++nd;
            }
            for (j = 0; j < ax->nloops; ++j) {
                d[nd].n = ax->loops[j].n; d[nd].is = lts[j]; d[nd].os = ax->loops[j].os;
                d[nd].tw = 0; d[nd].is_batch = (j == ax->batch_loop); ++nd;
            }
            flags = ax->flags_out & (FFTW_AMD_F_SWAP_OUT | FFTW_AMD_F_REAL_OUT);
            if (k == 2) flags |= FFTW_AMD_F_TW_IN;
            emit_pass(p, tmp, ax->dst, lens[k - 1], ts, Pf[k - 1] * ax->os, d, nd,
                      k == 2 ? ax->n : 0, flags);
        }
    }
    buf_release(p, tbuf);
}

/* Bluestein: y[k] = conj(w_k) * ( (x conj(w)) (*) w )[k], the convolution by
   two DFTs of the smooth size nb >= 2n-1  (reference A.c:1642-1688) */
static void emit_bluestein(plan *p, const fa_axis *ax) {
    i64 n = ax->n, nb = fa_next_smooth(2 * n - 1);
    i64 wts, lts[FA_MAXLOOPS], total;
    int chirp = tab_chirp(p, n, nb);
    int kern = tab_blue_kernel(p, n, nb);
    int wbuf, j, nd;
    fa_loc w;
    sdim d[FA_MAXLOOPS];
    fa_axis sub;

    /* work image [loops][nb], axis innermost */
    {
        fa_axis lay = *ax;
        lay.is = 1;   /* force the axis innermost in the scratch */
        total = scratch_layout(&lay, nb, &wts, lts);
    }
    wbuf = buf_acquire(p, total);
    w.buf = wbuf; w.base = 0; w.im = 1;

    nd = 0;
    for (j = 0; j < ax->nloops; ++j) {
        d[nd].n = ax->loops[j].n; d[nd].is = ax->loops[j].is; d[nd].os = lts[j];
        d[nd].tw = 0; d[nd].is_batch = (j == ax->batch_loop); ++nd;
    }
    emit_copy(p, FFTW_AMD_STEP_COPY, ax->src, w, nb, n, ax->is, wts, d, nd,
              (ax->flags_in & (FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_REAL_IN)) | FFTW_AMD_F_MUL_CONJ,
              chirp, -1);

    memset(&sub, 0, sizeof(sub));
    sub.n = nb; sub.is = wts; sub.os = wts; sub.src = w; sub.dst = w;
    sub.nloops = ax->nloops; sub.batch_loop = ax->batch_loop;
    for (j = 0; j < ax->nloops; ++j) {
        sub.loops[j].n = ax->loops[j].n; sub.loops[j].is = lts[j]; sub.loops[j].os = lts[j];
    }
    fa_emit_axis(p, &sub);

    for (j = 0; j < nd; ++j) d[j].is = d[j].os;
    emit_copy(p, FFTW_AMD_STEP_COPY, w, w, nb, nb, wts, wts, d, nd, FFTW_AMD_F_MUL_TABLE, kern, -1);

    sub.flags_in = FFTW_AMD_F_SWAP_IN;
    sub.flags_out = FFTW_AMD_F_SWAP_OUT;
    fa_emit_axis(p, &sub);

    for (j = 0; j < ax->nloops; ++j) { d[j].is = lts[j]; d[j].os = ax->loops[j].os; }
    emit_copy(p, FFTW_AMD_STEP_COPY, w, ax->dst, n, n, wts, ax->os, d, nd,
              (ax->flags_out & (FFTW_AMD_F_SWAP_OUT | FFTW_AMD_F_REAL_OUT)) | FFTW_AMD_F_MUL_CONJ,
              chirp, -1);
    buf_release(p, wbuf);
}

/* Rader for a prime p: a cyclic convolution of length p-1 over the
   multiplicative group (reference rader_apply A.c:4187-4261) */
static void emit_rader(plan *p, const fa_axis *ax) {
    i64 pr = ax->n, m = pr - 1;
    i64 wts, lts[FA_MAXLOOPS], xts, xlts[FA_MAXLOOPS], total, xtotal;
    int pf = tab_perm(p, pr, 0), pinv = tab_perm(p, pr, 1);
    int omega = tab_rader_kernel(p, pr);
    int wbuf, xbuf, j, nd;
    fa_loc w, x0;
    sdim d[FA_MAXLOOPS];
    fa_axis sub, lay;
    fftw_amd_step_desc *s;

    lay = *ax;
    lay.is = 1;
    total = scratch_layout(&lay, m, &wts, lts);
    xtotal = scratch_layout(&lay, 1, &xts, xlts);
    wbuf = buf_acquire(p, total);
    xbuf = buf_acquire(p, xtotal);
    w.buf = wbuf; w.base = 0; w.im = 1;
    x0.buf = xbuf; x0.base = 0; x0.im = 1;

    nd = 0;
    for (j = 0; j < ax->nloops; ++j) {
        d[nd].n = ax->loops[j].n; d[nd].is = ax->loops[j].is; d[nd].os = lts[j];
        d[nd].tw = 0; d[nd].is_batch = (j == ax->batch_loop); ++nd;
    }
    /* a[k] = x[g^k] */
    emit_copy(p, FFTW_AMD_STEP_COPY, ax->src, w, m, m, ax->is, wts, d, nd,
              (ax->flags_in & (FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_REAL_IN)) | FFTW_AMD_F_PERM_SRC,
              -1, pf);
    /* x0 */
    for (j = 0; j < nd; ++j) d[j].os = xlts[j];
    emit_copy(p, FFTW_AMD_STEP_COPY, ax->src, x0, 1, 1, ax->is, xts, d, nd,
              ax->flags_in & (FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_REAL_IN), -1, -1);

    memset(&sub, 0, sizeof(sub));
    sub.n = m; sub.is = wts; sub.os = wts; sub.src = w; sub.dst = w;
    sub.nloops = ax->nloops; sub.batch_loop = ax->batch_loop;
    for (j = 0; j < ax->nloops; ++j) {
        sub.loops[j].n = ax->loops[j].n; sub.loops[j].is = lts[j]; sub.loops[j].os = lts[j];
    }
    fa_emit_axis(p, &sub);

    /* pointwise product, DC fix-ups, Y[0].  The kernel walks vectors in the
       scratch order, which is the order of dims given here. */
    s = new_step(p, FFTW_AMD_STEP_RADER_MUL);
    s->src_buf = wbuf; s->src_base = 0; s->src_im = 1;
    s->dst_buf = ax->dst.buf; s->dst_base = ax->dst.base; s->dst_im = ax->dst.im;
    s->aux_buf = xbuf; s->aux_base = 0;
    s->aux_n = m;
    s->table = omega;
    s->flags = ax->flags_out & (FFTW_AMD_F_SWAP_OUT | FFTW_AMD_F_REAL_OUT);
    {
        /* dims sorted by scratch stride ascending so that the linear vector
           index equals the scratch vector index */
        sdim sd[FA_MAXLOOPS];
        int order[FA_MAXLOOPS], a, b, t;
        for (j = 0; j < ax->nloops; ++j) order[j] = j;
        for (a = 0; a < ax->nloops; ++a)
            for (b = a + 1; b < ax->nloops; ++b)
                if (lts[order[b]] < lts[order[a]]) { t = order[a]; order[a] = order[b]; order[b] = t; }
        for (j = 0; j < ax->nloops; ++j) {
            int o = order[j];
            sd[j].n = ax->loops[o].n; sd[j].is = lts[o]; sd[j].os = ax->loops[o].os;
            sd[j].tw = 0; sd[j].is_batch = (o == ax->batch_loop);
        }
        /* keep extent-1 dims out but the vector numbering intact: extent 1
           contributes a factor of 1 */
        step_set_dims(p, s, sd, ax->nloops, -1);
    }

    sub.flags_in = FFTW_AMD_F_SWAP_IN;
    sub.flags_out = FFTW_AMD_F_SWAP_OUT;
    fa_emit_axis(p, &sub);

    /* Y[g^-k] = c[k] */
    for (j = 0; j < ax->nloops; ++j) { d[j].is = lts[j]; d[j].os = ax->loops[j].os; }
    emit_copy(p, FFTW_AMD_STEP_COPY, w, ax->dst, m, m, wts, ax->os, d, nd,
              (ax->flags_out & (FFTW_AMD_F_SWAP_OUT | FFTW_AMD_F_REAL_OUT)) | FFTW_AMD_F_PERM_DST,
              -1, pinv);
    buf_release(p, xbuf);
    buf_release(p, wbuf);
}

/* lengths with a register kernel (pass1024 / passrr) */
static int has_register_kernel(i64 L) {
    return L == 1024 || (L <= 1024 && (fa_hip_rr_tile((int)L) > 0 || fa_hip_r3t_tile((int)L) > 0));
}

/* Three-pass split of a contiguous power of two (n >= 2^21).  The balanced split is not the
   fastest: per-pass times differ by position (first: long input stride; middle: in place at a
   medium stride; last: contiguous rows in, transposed out) and by tile width -- the 256-point
   kernel (16-wide tiles) is the slow one in the first two positions, short middle passes with
   64 ... 256-wide tiles the fast ones.  Costs: ms per 4 GiB moved, measured on MI355X
   (tests/perf_p512.py); unmeasured positions carry a pessimistic guess. */
static void pow2_three_pass_split(i64 n, i64 *lens) {
    /*                              2^4   2^5   2^6   2^7   2^8   2^9   2^10 */
    static const double c_first[] = { 1.00, 1.00, 0.95, 0.80, 1.02, 0.87, 1.30 };
    static const double c_mid[]   = { 0.77, 0.78, 0.75, 0.76, 0.99, 0.90, 1.02 };
    static const double c_last[]  = { 0.95, 0.95, 0.90, 0.80, 0.80, 0.82, 0.84 };
    int e = 0, a, m, c, ba = 0, bm = 0, bc = 0;
    double best = 1e30;
    while (((i64)1 << e) < n) ++e;
    for (a = 4; a <= 10; ++a)
        for (m = 4; m <= 10; ++m) {
            double t;
            c = e - a - m;
            if (c < 4 || c > 10) continue;
            t = c_first[a - 4] + c_mid[m - 4] + c_last[c - 4];
            {   /* near-ties go to the more balanced split */
                int hi = a > m ? (a > c ? a : c) : (m > c ? m : c), lo = a < m ? (a < c ? a : c) : (m < c ? m : c);
                t += 0.01 * (hi - lo);
            }
            if (t < best - 1e-9) { best = t; ba = a; bm = m; bc = c; }
        }
    if (!ba) return;
    lens[0] = (i64)1 << ba;
    lens[1] = (i64)1 << bm;
    lens[2] = (i64)1 << bc;
}

/* Three-pass split of a contiguous length that is not a power of two.  The balanced split the
   factoriser returns is rarely the fastest: the register kernels differ by up to 1.5x from one
   sub-transform length to the next (tile width, ragged tiles, radix mix) and by position (first:
   long input stride; middle: in place; last: rows in, transposed out).  Costs are measured
   medians, ms per GiB of a pass (split_costs.inc); the additive model picks the measured best
   split for 5 of the 8 lengths it was checked on and is within 1-7 % on the others
   (tools/perf/fit_split_costs.py).  Reference counterpart: the planner's choice among
   ct-dit radices by estimated cost (fftw/fftw_api.c:15300-15426). */
typedef struct { int L; double c[3]; } split_cost;
static const split_cost g_split_costs[] = {
#include "split_costs.inc"
};
static double split_cost_of(i64 L, int pos) {
    const int n = (int)(sizeof(g_split_costs) / sizeof(g_split_costs[0]));
    int i;
    for (i = 0; i < n - 1; ++i)
        if (g_split_costs[i].L == L && g_split_costs[i].c[pos] > 0.0) return g_split_costs[i].c[pos];
    return g_split_costs[n - 1].c[pos];
}
/* the measured value only: 0 when the table has no sample for (L, pos) */
static double split_cost_measured(i64 L, int pos) {
    const int n = (int)(sizeof(g_split_costs) / sizeof(g_split_costs[0]));
    int i;
    for (i = 0; i < n - 1; ++i)
        if (g_split_costs[i].L == L) return g_split_costs[i].c[pos];
    return 0.0;
}
/* factor on the first (row 0) / last (row 1) pass by min(5, v2(stride)): their strided side moves runs of
   T x 16 bytes that start at multiples of the stride, and runs that do not start on a 128-byte line cost up
   to 1.3x (tools/perf/fit_split_costs.py) */
static const double g_split_align[2][6] = {
#include "split_align.inc"
};
static int v2_capped(i64 x) { int k = 0; while (k < 5 && (x & 1) == 0) { x >>= 1; ++k; } return k; }

/* returns the modelled cost of the split it chose (ms per GiB of a pass, three passes), -1 if none.
   Candidates whose FIRST length is a multiple of 8 come first: with an odd or barely even first length the
   two later passes ran up to 1.3x slower than their lengths' medians in the sweeps (their loops over the first
   pass's output index then start off the 128-byte grid); restricted to L1 % 8 == 0 the model's pick is within
   4.3 % of the measured best on all ten lengths of the sweep, unrestricted it misses by up to 34 %. */
static double mixed_three_pass_split(i64 n, i64 *lens) {
    static const int need[4] = { 8, 4, 2, 1 };
    int level;
    for (level = 0; level < 4; ++level) {
        i64 a, c;
        double best = -1.0;
        for (a = 16; a <= 1024; ++a) {
            if (n % a || a % need[level] || !has_register_kernel(a)) continue;
            for (c = 16; c <= 1024; ++c) {
                i64 m, lo, hi;
                double cost;
                if ((n / a) % c || !has_register_kernel(c)) continue;
                m = n / a / c;
                if (m < 16 || m > 1024 || !has_register_kernel(m)) continue;
                lo = a < c ? a : c; if (m < lo) lo = m;
                hi = a > c ? a : c; if (m > hi) hi = m;
                if (hi > 4 * lo) continue;
                cost = split_cost_of(a, 0) * g_split_align[0][v2_capped(m * c)] + split_cost_of(m, 1) +
                       split_cost_of(c, 2) * g_split_align[1][v2_capped(a * m)];
                if (best < 0.0 || cost < best) { best = cost; lens[0] = a; lens[1] = m; lens[2] = c; }
            }
        }
        if (best >= 0.0) return best;
    }
    return -1.0;
}

/* Two-pass split n = L1 x L2 by the measured model of tools/perf/fit_split_costs.py (split2_costs.inc: median cost of
   every menu length as first / last pass over 919 timed plans of 78 lengths; split2_align.inc: the price of strided
   runs that start off the 128-byte grid, which depends on the OTHER length's factors of two and on the tile width T:
   cost = c * (1 + 8 / T * m[v2(other)])).  Against the measured best split: mean +0.8 %, worst +9.8 %; the balanced
   split it replaces: mean +9.6 %, worst +43 % (65 536 = 128 x 512 instead of 256 x 256: 2.67 vs 3.07 ms per 4 GiB). */
static const split_cost g_split2_costs[] = {
#include "split2_costs.inc"
};
static const double g_split2_align[2][6] = {
#include "split2_align.inc"
};
static double split2_cost_of(i64 L, int pos) {
    const int n = (int)(sizeof(g_split2_costs) / sizeof(g_split2_costs[0]));
    int i;
    for (i = 0; i < n - 1; ++i)
        if (g_split2_costs[i].L == L && g_split2_costs[i].c[pos] > 0.0) return g_split2_costs[i].c[pos];
    return g_split2_costs[n - 1].c[pos];
}
static int split_tile_of(i64 L) {
    int t = L == 1024 ? 8 : fa_hip_rr_tile((int)L);
    if (t <= 0) t = fa_hip_r3t_tile((int)L);
    return t > 0 ? t : 8;
}
static void mixed_two_pass_split(i64 n, i64 *lens) {
    i64 a;
    double best = -1.0;
    for (a = 16; a <= 1024; ++a) {
        i64 b;
        double cost;
        if (n % a || !has_register_kernel(a)) continue;
        b = n / a;
        if (b < 16 || b > 1024 || !has_register_kernel(b)) continue;
        if (a > 8 * b || b > 8 * a) continue;
        cost = split2_cost_of(a, 0) * (1.0 + 8.0 / split_tile_of(a) * g_split2_align[0][v2_capped(b)]) +
               split2_cost_of(b, 1) * (1.0 + 8.0 / split_tile_of(b) * g_split2_align[1][v2_capped(a)]);
        if (best < 0.0 || cost < best) { best = cost; lens[0] = a; lens[1] = b; }
    }
}

static int rows_alias_ok(const plan *p, const fa_axis *ax, fa_loc a, fa_loc b);

/* May a contiguous axis longer than FA_LMAX_SINGLE run as ONE trip of a register rows kernel?  Such a step has
   no LDS-kernel fallback (the row does not fit the runtime-radix kernel), so everything the rows kernel needs is
   settled here, at plan time: interleaved unit-stride rows, 16-byte aligned user arrays and even loop strides,
   rows that map onto themselves when the transform is in place, no two-level tile dim, no FFTW_UNALIGNED. */
static int long_rows_ok(const plan *p, const fa_axis *ax) {
    int j, rows = 0;
    if (ax->is != 2 || ax->os != 2 || ax->src.im != 1 || ax->dst.im != 1) return 0;
    if ((ax->flags_in | ax->flags_out) & ~(FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_SWAP_OUT)) return 0;
    if (!rows_alias_ok(p, ax, ax->src, ax->dst)) return 0;
    for (j = 0; j < ax->nloops; ++j) {
        if ((ax->loops[j].is % 2) || (ax->loops[j].os % 2)) return 0;
        if (ax->loops[j].n == 2 && (iabs(ax->loops[j].is) == 2 || iabs(ax->loops[j].os) == 2)) return 0;
        if (ax->loops[j].n > 1) rows = 1;
    }
    if (!rows) return 0;
    if ((ax->src.base % 2) || (ax->dst.base % 2)) return 0;
    if (ax->src.buf == 0 && ((size_t)p->ri % 16)) return 0;
    if (ax->src.buf == 1 && ((size_t)p->ro % 16)) return 0;
    if (ax->dst.buf == 0 && ((size_t)p->ri % 16)) return 0;
    if (ax->dst.buf == 1 && ((size_t)p->ro % 16)) return 0;
    return 1;
}

/* Bluestein for interleaved unit-stride rows whose padded length fits one workgroup: the whole algorithm in ONE
   kernel (pass3b.hpp).  The padded length comes from a ladder (blue_menu.inc); when its first entry >= 2n - 1 is
   more than 1.35 x that (short lengths: the ladder starts at 539) the step-by-step plan with its tighter
   nb = next smooth number is kept.  Like the long rows the step has no fallback, so the same plan-time conditions
   apply (long_rows_ok).  1 = emitted. */
static int emit_bluestein_rows(plan *p, const fa_axis *ax) {
    i64 need = 2 * ax->n - 1;
    int nb, j, nd = 0;
    sdim d[FA_MAXLOOPS];
    fftw_amd_step_desc *s;
    if (getenv("FFTW_AMD_NO_TUNED") || getenv("FFTW_AMD_NO_3S") || getenv("FFTW_AMD_NO_BLUE_ROWS")) return 0;
    if (ax->n < 2 || need > 16384 || ax->nloops == 0) return 0;
    nb = fa_hip_blue_nb((int)need);
    if (nb <= 0 || (double)nb > 1.35 * (double)need) return 0;
    if (!long_rows_ok(p, ax)) return 0;
    for (j = 0; j < ax->nloops; ++j) {
        d[nd].n = ax->loops[j].n; d[nd].is = ax->loops[j].is; d[nd].os = ax->loops[j].os;
        d[nd].tw = 0; d[nd].is_batch = (j == ax->batch_loop); ++nd;
    }
    emit_pass(p, ax->src, ax->dst, nb, 2, 2, d, nd, 0,
              (ax->flags_in & FFTW_AMD_F_SWAP_IN) | (ax->flags_out & FFTW_AMD_F_SWAP_OUT));
    if (p->failed) return 1;
    s = &p->steps[p->nsteps - 1];
    if (s->tile_lo_n != 1) { p->failed = 1; return 1; }
    s->variant = FFTW_AMD_K_BLUE;
    s->tile = fa_hip_blue_tile(nb);
    s->aux_n = ax->n;
    s->tw_lo = tab_chirp(p, ax->n, nb);
    s->tw_hi = tab_blue_kernel(p, ax->n, nb);
    /* emit_pass counted 5 nb log2 nb once; the kernel runs the stages twice plus three pointwise products */
    {
        double rows = 1.0;
        for (j = 0; j < ax->nloops; ++j) rows *= (double)ax->loops[j].n;
        p->est_flops += rows * (5.0 * (double)nb * log2((double)nb) + 18.0 * (double)nb);
    }
    return 1;
}

/* cost of a pass of L points on the 512-item strided / transposed three-stage kernels (r3tw_menu.inc), in ms per
   GiB moved like split_costs.inc: last = 0 first pass (columns), 1 last pass (rows in, transposed store, twiddle on
   the input).  Measured with tools/perf/perf_pow2_two_trip.py (profiles/r03_two_trip_wide.txt), taking the two
   2048-point passes as equal; 0: no such kernel */
static double wide_pass_cost(i64 L, int last) {
    static const struct { int L; double first, last; } c[] = {
        { 2048, 0.239, 0.239 }, { 2000, 0.247, 0.247 }, { 1920, 0.185, 0.207 }, { 1600, 0.265, 0.265 },
        { 1536, 0.274, 0.290 }, { 1440, 0.339, 0.339 }, { 1280, 0.227, 0.198 }, { 1200, 0.300, 0.300 },
        { 1080, 0.269, 0.298 },
    };
    size_t i;
    if (L <= 1024 || fa_hip_r3t_tile((int)L) < 8) return 0.0;
    for (i = 0; i < sizeof(c) / sizeof(c[0]); ++i)
        if (c[i].L == L) return last ? c[i].last : c[i].first;
    return 0.0;
}

static void fa_emit_axis(plan *p, const fa_axis *ax_in) {
    fa_axis ax = *ax_in;
    i64 lens[FA_MAXPASS];
    int k, contiguous;
    sdim d[FA_MAXLOOPS];
    i64 lmax1;

    if (p->failed) return;

    if (!fa_lds_able(ax.n)) {
        if (emit_bluestein_rows(p, &ax)) return;
        if (fa_is_prime(ax.n) && fa_lds_able(ax.n - 1)) emit_rader(p, &ax);
        else emit_bluestein(p, &ax);
        return;
    }
    /* A strided axis keeps tiles at least 8 wide so that global access stays
       in 128-byte segments; a contiguous axis may fill the tile by itself. */
    contiguous = (iabs(ax.is) <= 2 && iabs(ax.os) <= 2) || ax.dense;
    lmax1 = contiguous ? FA_LMAX_SINGLE : FA_TILE_ELEMS / 8;
    /* a strided axis with a register kernel of its own length needs no split either */
    if (!contiguous && ax.n <= 1024 && !getenv("FFTW_AMD_NO_TUNED") && has_register_kernel(ax.n)) lmax1 = 1024;
    /* ... and neither does a strided axis of 1025 ... 2048 points with a narrow-tile three-stage kernel
       (4 ... 7 sequences per tile, 64 ... 112-byte segments, ~3 TB/s): the 1080 of a 1080 x 1920 image */
    if (!contiguous && ax.n > 1024 && ax.n <= 2048 && ax.nloops > 0 && !getenv("FFTW_AMD_NO_TUNED") && !getenv("FFTW_AMD_NO_NARROW") &&
        fa_hip_r3t_tile((int)ax.n) > 0 && ax.src.im == 1 && ax.dst.im == 1 &&
        !((ax.flags_in | ax.flags_out) & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT)))
        lmax1 = 2048;
    /* powers of two above 1024 run faster as two register-kernel passes than
       as one LDS-sized pass (measured: 4096-point rows 0.8 TB/s vs ~5 TB/s per pass) */
    if (contiguous && ax.n > 1024 && (ax.n & (ax.n - 1)) == 0 && ax.nloops > 0 &&
        !getenv("FFTW_AMD_NO_TUNED")) {
        /* ... except contiguous rows of 2048 / 4096: the three-stage kernel does them in one */
        int rows3s = fa_hip_r3_tile((int)ax.n) > 0 && iabs(ax.is) == 2 && iabs(ax.os) == 2 &&
                     ax.src.im == 1 && ax.dst.im == 1 &&
                     !((ax.flags_in | ax.flags_out) & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT)) &&
                     !getenv("FFTW_AMD_NO_3S");
        /* 8192: one row per workgroup (32 x 16 x 16), one trip instead of 128 x 64 in two; 16384: one row per
           512-item workgroup (pass3w.hpp) */
        if (rows3s && ax.n > FA_LMAX_SINGLE) rows3s = ax.n <= 16384 && long_rows_ok(p, &ax);
        if (!rows3s) lmax1 = 1024;
        else if (ax.n > lmax1) lmax1 = ax.n;
    }
    if (ax.nloops == 0 && !contiguous) lmax1 = FA_LMAX_SINGLE;
    /* 1024 < n <= 4096 that is not a power of two: two register-kernel passes (each near
       the copy rate) beat one pass of the LDS kernel (1.3-1.8 TB/s) whenever both halves of
       a balanced split have a register kernel -- measured n = 3000: 1.27 vs 2.4 TB/s */
    /* 4096 < n < 16384 with a three-stage rows kernel (one row per workgroup; above 8192 the 512-item kernels of
       kernels_r3w.hip): one trip instead of two, under the same plan-time conditions as the 8192-point rows (no
       LDS-kernel fallback at that length) */
    if (contiguous && ax.n > FA_LMAX_SINGLE && ax.n < 16384 && (ax.n & (ax.n - 1)) != 0 && ax.nloops > 0 &&
        !getenv("FFTW_AMD_NO_TUNED") && !getenv("FFTW_AMD_NO_3S") && fa_hip_r3_tile((int)ax.n) > 0 &&
        long_rows_ok(p, &ax))
        lmax1 = ax.n;
    if (contiguous && ax.n > 1024 && ax.n <= lmax1 && (ax.n & (ax.n - 1)) != 0 && ax.nloops > 0 &&
        !getenv("FFTW_AMD_NO_TUNED") &&
        /* ... unless the three-stage rows kernel does the whole length in one trip */
        !(fa_hip_r3_tile((int)ax.n) > 0 && iabs(ax.is) == 2 && iabs(ax.os) == 2 && ax.src.im == 1 &&
          ax.dst.im == 1 && !((ax.flags_in | ax.flags_out) & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT)) &&
          !getenv("FFTW_AMD_NO_3S"))) {
        i64 l2[FA_MAXPASS];
        if (fa_factor_passes_pref(ax.n, 2, 1, 1024, l2, has_register_kernel) == 2 &&
            has_register_kernel(l2[0]) && has_register_kernel(l2[1]))
            lmax1 = 1024;
    }
    k = fa_factor_passes_pref(ax.n, FA_MAXPASS, lmax1, contiguous ? p->cfg.lmax_multi : FA_TILE_ELEMS / 8, lens,
                              getenv("FFTW_AMD_NO_TUNED") ? NULL : has_register_kernel);
    /* n = L1 x L2 with 1024 < L1 <= 2048 in TWO trips instead of three: the strided three-stage kernel of
       such a length works on tiles of 4 ... 7 columns (64 ... 112-byte segments, ~3.2 TB/s) -- slower per trip
       than the 128-byte kernels, but two trips beat three (2^21 = 2048 x 1024: 7.9 vs 11.2 ms per 8 GiB; with
       BOTH lengths above 1024 the two slow trips only tie with three fast ones, so that is not done) */
    /* (the narrow-tile kernel takes interleaved complex data on 16-byte aligned arrays only -- the same
       conditions as for a strided axis above; with split or unaligned arrays its pass would fall to the
       runtime-radix LDS kernel with a 1 ... 2 column tile, far slower than the three-pass plan) */
    if (k == 3 && contiguous && ax.nloops > 0 && !getenv("FFTW_AMD_NO_TUNED") && !getenv("FFTW_AMD_NO_NARROW") &&
        ax.src.im == 1 && ax.dst.im == 1 && !(p->flags & FFTW_UNALIGNED) && p->cfg.lmax_multi >= 1024 &&
        !((ax.flags_in | ax.flags_out) & (FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT))) {
        /* cost per GiB of a pass (the units of split_costs.inc): narrow-tile first pass 0.34 (2048-point: 5.33 ms
           per 16 GiB), the partner as LAST pass from the table (1024: 0.18, the cfg2 pass; 1000: 0.20); the
           three-pass alternative from the same table (powers of two: 0.68, measured 2^21 ... 2^24) */
        i64 L1, bestL1 = 0, l3[FA_MAXPASS];
        double best2 = 1e30, est3 = 0.68;
        if ((ax.n & (ax.n - 1)) != 0) {
            double c3 = mixed_three_pass_split(ax.n, l3);
            if (c3 > 0.0) est3 = c3;
        }
        for (L1 = 2048; L1 > 1024; --L1) {
            i64 L2 = ax.n / L1;
            double c1, c2;
            if (ax.n % L1 || fa_hip_r3t_tile((int)L1) <= 0 || L2 < 96) continue;
            /* round 3: the 512-item kernels of r3tw_menu.inc (8 ... 15 sequences per tile) are cheaper first passes
               than the narrow tiles, and serve as LAST pass too (rows in, transposed store, twiddle on the input):
               both lengths above 1024, e.g. 2^22 = 2048 x 2048 in 3.8 instead of 5.4 ms per 4 GiB */
            c1 = wide_pass_cost(L1, 0);
            if (c1 <= 0.0) c1 = 0.34;
            if (L2 > 1024) {
                c2 = L2 <= 2048 ? wide_pass_cost(L2, 1) : 0.0;
                if (c2 <= 0.0 || wide_pass_cost(L1, 0) <= 0.0) continue;
            } else {
                if (!has_register_kernel(L2)) continue;
                c2 = L2 == 1024 ? 0.18 : (L2 == 1000 ? 0.20 : split_cost_measured(L2, 2));
            }
            if (c2 <= 0.0) continue;                  /* no measurement for that partner: do not guess */
            if (c1 + c2 < best2) { best2 = c1 + c2; bestL1 = L1; }
        }
        if (bestL1 && best2 < 0.95 * est3) { k = 2; lens[0] = bestL1; lens[1] = ax.n / bestL1; }
    }
    /* the measured split models know lengths up to 1024: with a smaller cap (the FFTW_MEASURE candidate
       lmax_multi = 512, FFTW_AMD_LMAX_MULTI) the balanced split of fa_factor_passes_pref, which honours it, stays */
    if (p->cfg.lmax_multi < 1024) { /* keep lens */ }
    else if (k == 3 && contiguous && (ax.n & (ax.n - 1)) == 0 && !getenv("FFTW_AMD_NO_TUNED")) pow2_three_pass_split(ax.n, lens);
    else if (k == 3 && contiguous && !getenv("FFTW_AMD_NO_TUNED") && !getenv("FFTW_AMD_NO_SPLIT_COSTS")) mixed_three_pass_split(ax.n, lens);
    else if (k == 2 && contiguous && ax.nloops > 0 && lens[0] <= 1024 && lens[1] <= 1024 && !getenv("FFTW_AMD_NO_TUNED") &&
             !getenv("FFTW_AMD_NO_SPLIT_COSTS")) mixed_two_pass_split(ax.n, lens);
    {
        /* test hook: FFTW_AMD_FORCE_LENS="L1,L2,..." fixes the split of the axis whose length
           is the product (tests/test_gpu_menu.py reaches every kernel variant with it) */
        const char *e = getenv("FFTW_AMD_FORCE_LENS");
        if (e && *e) {
            i64 fl[FA_MAXPASS], prod = 1;
            int fk = 0;
            const char *c = e;
            while (*c && fk < FA_MAXPASS) {
                char *end;
                long long v = strtoll(c, &end, 10);
                if (end == c || v < 1) { fk = 0; break; }
                fl[fk++] = v;
                prod *= v;
                c = (*end == ',') ? end + 1 : end;
                if (*end && *end != ',') { fk = 0; break; }
            }
            if (fk >= 2 && prod == ax.n) {
                int a;
                k = fk;
                for (a = 0; a < fk; ++a) lens[a] = fl[a];
            }
        }
    }
    if (k == 0) {
        /* e.g. a prime factor between lmax and FA_PRIME_LDS_MAX cannot happen;
           sizes that do not split fall back to Bluestein */
        emit_bluestein(p, &ax);
        return;
    }
    if (k == 1) {
        int nd = loops_to_sdims(&ax, d, 0);
        emit_pass(p, ax.src, ax.dst, ax.n, ax.is, ax.os, d, nd, 0,
                  (ax.flags_in & (FFTW_AMD_F_SWAP_IN | FFTW_AMD_F_REAL_IN)) |
                  (ax.flags_out & (FFTW_AMD_F_SWAP_OUT | FFTW_AMD_F_REAL_OUT)));
        return;
    }
    if (ax.nloops + k > FFTW_AMD_MAX_DIMS) { p->failed = 1; return; }
    if (p->cfg.long_first) {
        /* longest sub-transform first (FFTW_MEASURE candidate) */
        int a, b;
        for (a = 0; a < k; ++a)
            for (b = a + 1; b < k; ++b)
                if (lens[b] > lens[a]) { i64 t = lens[a]; lens[a] = lens[b]; lens[b] = t; }
    }
    emit_ct(p, &ax, lens, k);
}

/* ------------------------------------------------------- whole problems */

/* merge loop b into loop a when b is exactly a's inner continuation */
static int merge_loops(fa_dim *l, int n, int *batch) {
    int changed = 1;
    while (changed) {
        int a, b;
        changed = 0;
        for (a = 0; a < n && !changed; ++a)
            for (b = 0; b < n && !changed; ++b) {
                if (a == b || a == *batch || b == *batch) continue;
                if (l[a].is == l[b].n * l[b].is && l[a].os == l[b].n * l[b].os) {
                    int i;
                    l[b].n *= l[a].n;
                    for (i = a; i + 1 < n; ++i) l[i] = l[i + 1];
                    if (*batch > a) --*batch;
                    --n;
                    changed = 1;
                }
            }
    }
    return n;
}

/* loops for transforming `axis` of a rank-r array: the other transform dims
   plus the howmany dims.  use_os: source strides are the output strides. */
static int collect_loops(const plan *p, const fa_dim *dims, int rank, int axis,
                         const i64 *ext_override, int use_os, fa_axis *ax) {
    int i, n = 0;
    ax->batch_loop = -1;
    for (i = 0; i < rank; ++i) {
        if (i == axis) continue;
        if (n >= FA_MAXLOOPS) return -1;
        ax->loops[n].n = ext_override ? ext_override[i] : dims[i].n;
        ax->loops[n].is = use_os ? dims[i].os : dims[i].is;
        ax->loops[n].os = dims[i].os;
        ++n;
    }
    for (i = 0; i < p->hrank; ++i) {
        if (n >= FA_MAXLOOPS) return -1;
        ax->loops[n].n = (i == 0) ? p->chunk : p->hdims[i].n;
        ax->loops[n].is = use_os ? p->hdims[i].os : p->hdims[i].is;
        ax->loops[n].os = p->hdims[i].os;
        if (i == 0) ax->batch_loop = n;
        ++n;
    }
    n = merge_loops(ax->loops, n, &ax->batch_loop);
    ax->nloops = n;
    return 0;
}


/* A two-dimensional transform n0 x n1 whose rows (n1 = 2048 or 4096 points, contiguous) have a one-trip rows
   kernel holding T whole rows per tile, and whose strided axis is n0 = T x L0 with 1024 < n0 <= 4096 and L0 <= 1024
   a register-kernel length, in TWO full-rate trips (pass3q.hpp, pass3s.hpp XROW): the strided axis is split
   Cooley-Tukey over the row index r = s + T r' --
     trip 1  the ordinary strided L0-point pass of every row class s (128-byte segments), twiddle w_n0^(s k') on
             the output, stored as [k'][s][c]: the T rows that belong together are adjacent
     trip 2  tiles k' of T rows: the row transforms and the DFT of length T across the rows in the same registers,
             X[k' + L0 q] = sum_s w_T^(s q) DFT_n1(Y_s[k'])
   Round 2 ran 4096 x 4096 as rows + 64 x 64 columns (three trips, 23 %) and strided axes of 1025 ... 2048 points on
   the narrow-tile kernel (64-byte segments, 3.2 TB/s).  The last step has no other executor, so everything it
   needs is settled here: interleaved unit-stride rows, 16-byte aligned arrays, even strides, at most one batch
   loop, no FFTW_UNALIGNED.  1 = emitted. */
static int has_register_kernel(i64 L);
static int emit_rows_lo_dft(plan *p, fa_loc in, fa_loc out, int sw_in, int sw_out) {
    const fa_dim *col = &p->dims[0], *row = &p->dims[1];
    sdim d[3];
    int nd, sbuf;
    i64 N1, N0, T = 0, L0, img, chunk = p->hrank ? p->chunk : 1;
    fftw_amd_step_desc *st;
    if (p->rank != 2 || p->hrank > 1 || getenv("FFTW_AMD_NO_TUNED") || getenv("FFTW_AMD_NO_LO_DFT") || getenv("FFTW_AMD_NO_3S")) return 0;
    N1 = row->n; N0 = col->n;
    if ((N1 != 2048 && N1 != 4096) || N0 <= 1024 || N0 > 4096) return 0;
    if (N1 == 4096 && N0 <= 2048 && N0 % 2 == 0 && has_register_kernel(N0 / 2)) T = 2;      /* pass3s<16>, XROW */
    else if (N0 % 4 == 0 && has_register_kernel(N0 / 4)) T = 4;                             /* pass3q / pass3s<8>, XROW */
    if (!T) return 0;
    L0 = N0 / T;
    if (row->is != 2 || row->os != 2 || in.im != 1 || out.im != 1) return 0;
    if (col->is < 2 * N1 || col->os < 2 * N1 || (col->is % 2) || (col->os % 2)) return 0;
    if (p->flags & FFTW_UNALIGNED) return 0;
    if (((size_t)p->ri % 16) || ((size_t)p->ro % 16)) return 0;
    if (p->hrank && ((p->hdims[0].is % 2) || (p->hdims[0].os % 2))) return 0;
    if (fa_hip_r3_tile((int)N1) <= 0) return 0;
    img = N0 * N1 * 2;
    sbuf = buf_acquire(p, chunk * img);
    {
        /* trip 1: L0-point column pass of every row class s, twiddle w_n0^(s k') on the output */
        fa_loc scr = { sbuf, 0, 1 };
        d[0].n = N1; d[0].is = 2;       d[0].os = 2;      d[0].tw = 0; d[0].is_batch = 0;    /* columns c */
        d[1].n = T;  d[1].is = col->is; d[1].os = 2 * N1; d[1].tw = 1; d[1].is_batch = 0;    /* row class s */
        nd = 2;
        if (p->hrank) { d[2].n = chunk; d[2].is = p->hdims[0].is; d[2].os = img; d[2].tw = 0; d[2].is_batch = 1; nd = 3; }
        emit_pass(p, in, scr, L0, T * col->is, T * 2 * N1, d, nd, N0, sw_in);
        if (p->failed) return 1;
        /* trip 2: tiles k' of T rows: DFT-n1 along every row and DFT-T across them */
        d[0].n = L0; d[0].is = T * 2 * N1; d[0].os = col->os; d[0].tw = 0; d[0].is_batch = 0;
        nd = 1;
        if (p->hrank) { d[1].n = chunk; d[1].is = img; d[1].os = p->hdims[0].os; d[1].tw = 0; d[1].is_batch = 1; nd = 2; }
        emit_pass(p, scr, out, N1, 2, 2, d, nd, 0, sw_out);
        if (p->failed) return 1;
        st = &p->steps[p->nsteps - 1];
        st->tile_lo_n = (int)T;
        st->tile_lo_is = 2 * N1;
        st->tile_lo_os = L0 * col->os;
        st->flags |= FFTW_AMD_F_LO_DFT;
        st->variant = FFTW_AMD_K_R3;
        st->tile = (int)T;
        p->est_flops += 5.0 * (double)(N0 * N1) * (double)chunk * (T == 4 ? 2.0 : 1.0);     /* the radix-T stage */
    }
    buf_release(p, sbuf);
    return 1;
}

static void build_c2c_on(plan *p, fa_loc in, fa_loc out);

static void build_c2c(plan *p) {
    fa_loc in = { 0, 0, p->in_im }, out = { 1, 0, p->out_im };
    if (p->via_scratch) {
        /* in-place with different strides on the two sides (same locations): transform into a dense row-major
           scratch image [hdims...][dims...], then copy it to the output layout -- every read of the user's array
           precedes every write (single chunk).  The reference does these problems with its in-place square
           DIF + transpose codelets or with buffers (A.c:2204-2250, 2596-2727); see api.c inplace_same_locations. */
        fa_dim sd[FA_MAXRANK], sh[FA_MAXRANK];
        fa_loc scr;
        fa_axis ax;
        sdim d[FA_MAXLOOPS];
        i64 stride = 2, total;
        int i, nd, sbuf;
        memcpy(sd, p->dims, sizeof(sd));
        memcpy(sh, p->hdims, sizeof(sh));
        for (i = p->rank - 1; i >= 0; --i) { p->dims[i].os = stride; stride *= p->dims[i].n; }
        for (i = p->hrank - 1; i >= 0; --i) {
            p->hdims[i].os = stride;
            stride *= (i == 0 ? p->chunk : p->hdims[i].n);
        }
        total = stride;
        sbuf = buf_acquire(p, total);
        scr.buf = sbuf; scr.base = 0; scr.im = 1;
        build_c2c_on(p, in, scr);
        /* the copy: all dims as loops, source = dense strides, destination = the user's output strides */
        memset(&ax, 0, sizeof(ax));
        for (i = 0; i < p->rank; ++i) p->dims[i].is = p->dims[i].os;
        for (i = 0; i < p->hrank; ++i) p->hdims[i].is = p->hdims[i].os;
        for (i = 0; i < p->rank; ++i) p->dims[i].os = sd[i].os;
        for (i = 0; i < p->hrank; ++i) p->hdims[i].os = sh[i].os;
        if (!p->failed && collect_loops(p, p->dims, p->rank, -1, NULL, 0, &ax) == 0) {
            nd = loops_to_sdims(&ax, d, 0);
            emit_pass(p, scr, out, 1, 0, 0, d, nd, 0, 0);
        } else p->failed = 1;
        memcpy(p->dims, sd, sizeof(sd));
        memcpy(p->hdims, sh, sizeof(sh));
        buf_release(p, sbuf);
        return;
    }
    build_c2c_on(p, in, out);
}

static void build_c2c_on(plan *p, fa_loc in, fa_loc out) {
    int a, first = 1;
    int sw_in = (p->sign > 0) ? FFTW_AMD_F_SWAP_IN : 0;
    int sw_out = (p->sign > 0) ? FFTW_AMD_F_SWAP_OUT : 0;
    if (p->rank == 0) {
        /* rank-0 "transform" = strided copy (reference rank0 solvers) */
        fa_axis ax;
        sdim d[FA_MAXLOOPS];
        int nd;
        memset(&ax, 0, sizeof(ax));
        if (collect_loops(p, p->dims, 0, -1, NULL, 0, &ax)) { p->failed = 1; return; }
        nd = loops_to_sdims(&ax, d, 0);
        emit_pass(p, in, out, 1, 0, 0, d, nd, 0, 0);
        return;
    }
    if (emit_rows_lo_dft(p, in, out, sw_in, sw_out)) return;
    for (a = p->rank - 1; a >= 0; --a) {
        fa_axis ax;
        memset(&ax, 0, sizeof(ax));
        ax.n = p->dims[a].n;
        ax.is = first ? p->dims[a].is : p->dims[a].os;
        ax.os = p->dims[a].os;
        ax.src = first ? in : out;
        ax.dst = out;
        ax.flags_in = sw_in;
        ax.flags_out = sw_out;
        if (collect_loops(p, p->dims, p->rank, a, NULL, !first, &ax)) { p->failed = 1; return; }
        fa_emit_axis(p, &ax);
        first = 0;
    }
}

/* passes a contiguous axis of length n needs (mirrors fa_emit_axis); 99 for
   lengths that go through Rader / Bluestein */
static int axis_pass_count(const plan *p, i64 n) {
    i64 lens[FA_MAXPASS], lmax1 = FA_LMAX_SINGLE;
    int k;
    if (!fa_lds_able(n)) return 99;
    /* (the radix-4 real plans feed strided pair data: the 3-stage row kernel does not apply) */
    if (n > 1024 && (n & (n - 1)) == 0 && !getenv("FFTW_AMD_NO_TUNED")) lmax1 = 1024;
    k = fa_factor_passes(n, FA_MAXPASS, lmax1, p->cfg.lmax_multi, lens);
    return k ? k : 99;
}

/* The quarter-length transforms of the radix-4 real plans run on PAIRS of interleaved vectors (a two-level tile dim),
   which only the power-of-two two-stage kernels and the 1024-point kernel take; a multi-pass m with another factor
   would put a pass on the runtime-radix LDS kernel (3 932 160 = 4 x 960 x 1024: 7.9 ms per 4 GiB where the half-length
   plan takes 4.3, profiles/r03_r2c_two_trip.txt).  One-pass lengths keep the old rule. */
static int radix4_passes_have_kernels(const plan *p, i64 m) {
    return axis_pass_count(p, m) <= 1 || (m & (m - 1)) == 0;
}

/* passes of the half-length complex transform of an r2c / c2r axis: its rows are contiguous, so a length with
   a three-stage rows kernel (up to 8192) is one pass */
static int half_axis_pass_count(const plan *p, i64 n) {
    if (n <= 16384 && !getenv("FFTW_AMD_NO_TUNED") && !getenv("FFTW_AMD_NO_3S") && fa_hip_r3_tile((int)n) > 0) return 1;
    return axis_pass_count(p, n);
}

/* r2c: last dim real -> half spectrum, then complex DFTs over the other dims
   on the half-spectrum array (reference rank_geq2_rdft2 A.c:10111-10282).
   p->dims[].is are strides of the REAL array (doubles), .os of the complex
   array (doubles, so 2 per complex). */
/* real -> half spectrum along one axis.  ax: the loops (.is = strides in the real
   source, .os = strides in the complex destination) and the batch loop; rs / cs:
   element strides of the transform index on the two sides, in doubles. */
/* the fused rows kernel loads 16-byte pairs and stores 16-byte complex numbers: every
   loop stride must keep that alignment, the user arrays must be 16-byte aligned, there must
   be a loop to tile over, and the tile dim must not be a two-level (pair) dim */
/* Do source and destination of a one-trip rows step alias?  The fused rows kernels read a tile
   of rows and write the same tile's results without a scratch image in between, so when both
   sides are the same user memory every row must map onto itself (FFTW's padded in-place
   layout does); otherwise the general path, which reads a whole chunk into scratch first. */
static int rows_alias_ok(const plan *p, const fa_axis *ax, fa_loc a, fa_loc b) {
    int j, same_mem;
    /* FFTW_UNALIGNED: the plan must serve arrays of any alignment through the new-array
       interface; the rows kernels rely on 16-byte alignment and have no fallback */
    if (p->flags & FFTW_UNALIGNED) return 0;
    same_mem = (a.buf == b.buf && a.buf < 2) ||
                      (a.buf < 2 && b.buf < 2 && (void *)p->ri == (void *)p->ro && p->ri != NULL);
    if (!same_mem) return 1;
    if (a.base != b.base) return 0;
    for (j = 0; j < ax->nloops; ++j)
        if (ax->loops[j].n > 1 && ax->loops[j].is != ax->loops[j].os) return 0;
    return 1;
}

static int r2c_rows_layout_ok(const plan *p, const fa_axis *ax, fa_loc in, fa_loc out, int epi, int pre) {
    int j, rows = 0;
    if (getenv("FFTW_AMD_NO_R2CROWS")) return 0;
    if (!rows_alias_ok(p, ax, in, out)) return 0;
    for (j = 0; j < ax->nloops; ++j) {
        if ((!pre && (ax->loops[j].is % 2)) || (!epi && (ax->loops[j].os % 2))) return 0;
        /* emit_pass would fold such a loop into a two-level tile dim, which this kernel lacks */
        if (ax->loops[j].n == 2 && (iabs(ax->loops[j].is) == 2 || iabs(ax->loops[j].os) == 2)) return 0;
        if (ax->loops[j].n > 1) rows = 1;
    }
    if (!rows) return 0;
    if (!pre && in.buf == 0 && (((size_t)p->ri % 16) || (in.base % 2))) return 0;
    if (!epi && out.buf == 1 && (((size_t)p->ro % 16) || (out.base % 2))) return 0;
    if (!pre && in.buf == 1 && (((size_t)p->ro % 16) || (in.base % 2))) return 0;
    return 1;
}

static int c2r_rows_layout_ok(const plan *p, const fa_axis *ax, fa_loc cur, fa_loc out, int pro, int post) {
    int j, rows = 0;
    if (getenv("FFTW_AMD_NO_R2CROWS")) return 0;
    if (!rows_alias_ok(p, ax, cur, out)) return 0;
    for (j = 0; j < ax->nloops; ++j) {
        if ((!pro && (ax->loops[j].is % 2)) || (!post && (ax->loops[j].os % 2))) return 0;
        if (ax->loops[j].n == 2 && (iabs(ax->loops[j].is) == 2 || iabs(ax->loops[j].os) == 2)) return 0;
        if (ax->loops[j].n > 1) rows = 1;
    }
    if (!rows) return 0;
    if (!pro && cur.buf == 0 && (((size_t)p->ri % 16) || (cur.base % 2))) return 0;
    if (!pro && cur.buf == 1 && (((size_t)p->ro % 16) || (cur.base % 2))) return 0;
    if (!post && out.buf == 1 && (((size_t)p->ro % 16) || (out.base % 2))) return 0;
    return 1;
}

/* rows per tile of the one-trip real-rows kernels for half length `half` (0: none): the power-of-two two-stage
   form carries the r2r hooks; its mixed-radix lengths (r2cr_menu.inc) and the three-stage form (640 ... 8192) only
   the plain r2c / c2r */
static int real_rows_tile(i64 half, int hooks) {
    int t;
    if (half > 16384) return 0;
    t = fa_hip_r2c_rows_tile((int)half);
    if (t > 0) return t;
    if (hooks) return 0;
    t = fa_hip_r2c_rows2m_tile((int)half);
    if (t > 0) return t;
    if (getenv("FFTW_AMD_NO_3S")) return 0;
    return fa_hip_r2c_rows3_tile((int)half);
}

/* dense real rows of n = 2 half = 4 ... 64 points with at least one tile of them in a loop whose strides are exactly
   the row lengths (n reals, half + 1 complex numbers): the one-stage real-rows kernel (pass1r.hpp); 0: not this */
static int short_real_rows_tile(const fa_axis *ax, i64 half, int hooks, int fwd) {
    int j, t;
    if (hooks || half < 2 || half > 32 || getenv("FFTW_AMD_NO_R1")) return 0;
    t = fa_hip_r2c_rows1_tile((int)half);
    if (t <= 0) return 0;
    for (j = 0; j < ax->nloops; ++j)
        if (ax->loops[j].n >= 256 && ax->loops[j].is == (fwd ? 2 * half : 2 * (half + 1)) &&
            ax->loops[j].os == (fwd ? 2 * (half + 1) : 2 * half)) return t;
    /* FFTW's padded rows (the in-place layout): 2 (half + 1) doubles per row on both sides */
    for (j = 0; j < ax->nloops; ++j)
        if (ax->loops[j].n >= 256 && ax->loops[j].is == 2 * (half + 1) && ax->loops[j].os == 2 * (half + 1)) return t;
    return 0;
}

/* can the r2c / c2r axis emitters fuse an r2r epilogue / prologue for this length? */
static int r2r_can_fuse(i64 nl) { return nl >= 2 && nl % 2 == 0 && !getenv("FFTW_AMD_R2R_UNFUSED"); }

/* twiddle tables of an untangle / tangle step.  Plain: modulus nl.  With a fused
   REDFT/RODFT 10/01 epilogue or prologue: the modulus-4n table serves both the
   r2r twiddle w_4n^k and the untangle twiddle w_n^k = w_4n^(4k) (step.tile = 4 is
   the index multiplier of the latter). */
static void r2r_fuse_tables(plan *p, fftw_amd_step_desc *s, i64 nl, int mode) {
    s->variant = mode;
    if (mode == FFTW_AMD_R2R_POST_E10 || mode == FFTW_AMD_R2R_POST_O10 ||
        mode == FFTW_AMD_R2R_PRE_E01 || mode == FFTW_AMD_R2R_PRE_O01) {
        tab_tw2(p, 4 * nl, &s->tw_lo, &s->tw_hi, &s->tw_shift);
        s->tile = 4;
    } else {
        tab_tw2(p, nl, &s->tw_lo, &s->tw_hi, &s->tw_shift);
        s->tile = 1;
    }
}

/* A long real transform in TWO trips (round 3): n = L1 x L2 real points, decimated over the real data itself
   instead of packed into a half-length complex transform plus an untangle trip.
     trip 1   the plain complex pass of length L1 down the columns of the input read as [L1][L2 / 2] pairs
              (x[j1 L2 + 2c], x[j1 L2 + 2c + 1]) -> Z[k1][c] in scratch: the real DFTs of two columns in one
     trip 2   rows k1 = 0 ... L1 / 2: separate the two columns while loading (rows k1 and L1 - k1 of Z), twiddle,
              DFT of length L2 along the row, store X[k1 + L1 k2] and, conjugated, X[n - (k1 + L1 k2)]
              (FFTW_AMD_F_REAL_DEC, pass3t_kernel RD = 1)
   Both trips move n / 2 complex numbers in and out: 2 x the algorithmic bytes where the half-length plans need
   3 x (two passes + untangle).  2^22 = 2048 x 2048 (BASELINE cfg3), 2^21 = 1024 x 2048.
   MEASURED (tools/perf/perf_r2c_two_trip.py, profiles/r03_r2c_two_trip.txt): with the 512-item kernels (one
   workgroup per CU, ~4.5 TB/s per trip with two chunk lanes) the two trips only TIE with the three trips of the
   256-item kernels, whose scratch chunks stay in the Infinity Cache (2^22: 4.58 against 4.11 ms per 4 GiB; 2^20:
   3.98 against 4.10) -- so the plan is off under FFTW_ESTIMATE (cfg.real_dec, FFTW_AMD_REAL_DEC=1) and a candidate
   of the FFTW_MEASURE search for r2c problems.  Reference counterpart: the
   rdft2 Cooley-Tukey plan over real-data codelets (ct_hc2c_direct_apply, fftw/fftw_api.c:5831, with
   fftw/rdft_scalar/r2cf/hc2cfdft_*.c).  The last step has no fallback executor, so everything it needs is settled
   here (aligned interleaved arrays, even strides).  1 = emitted. */
static int emit_r2c_decimated(plan *p, i64 nl, const fa_axis *axp, fa_loc in, fa_loc out) {
    fa_axis ax = *axp, c_ax, lay;
    const i64 L2 = 2048;
    i64 L1, h = nl / 2, zts, lts[FA_MAXLOOPS + 1], total;
    int zbuf, j, nd, cloop, n0 = p->nsteps;
    double flops0 = p->est_flops;
    fa_loc z;
    sdim d[FA_MAXLOOPS + 1];
    fftw_amd_step_desc *s;
    if (!p->cfg.real_dec || getenv("FFTW_AMD_NO_TUNED") || getenv("FFTW_AMD_NO_NARROW") ||
        (p->flags & FFTW_UNALIGNED) || p->cfg.lmax_multi < 1024)
        return 0;
    if (nl % L2 || !fa_hip_r3tw_rdec((int)L2) || fa_hip_r3t_tile((int)L2) < 8) return 0;
    L1 = nl / L2;
    if (L1 % 2 || L1 < 256) return 0;
    /* trip 1 must be ONE strided pass: a register kernel of that length with 128-byte segments */
    if (!((L1 <= 1024 && has_register_kernel(L1)) || (L1 <= 2048 && fa_hip_r3t_tile((int)L1) >= 8))) return 0;
    if (ax.nloops >= FA_MAXLOOPS || in.im != 0 || out.im != 1) return 0;
    if ((in.base % 2) || (out.base % 2) || ((size_t)p->ri % 16) || ((size_t)p->ro % 16)) return 0;
    for (j = 0; j < ax.nloops; ++j)
        if ((ax.loops[j].is % 2) || (ax.loops[j].os % 2)) return 0;

    lay = ax;
    lay.is = 1;
    total = scratch_layout(&lay, h, &zts, lts);
    zbuf = buf_acquire(p, total);
    z.buf = zbuf; z.base = 0; z.im = 1;

    c_ax = ax;
    cloop = c_ax.nloops++;
    c_ax.loops[cloop].n = L2 / 2;
    c_ax.loops[cloop].is = 2;
    c_ax.loops[cloop].os = zts;
    for (j = 0; j < ax.nloops; ++j) c_ax.loops[j].os = lts[j];
    c_ax.n = L1;
    c_ax.is = L2;                      /* L2 / 2 pairs of two doubles */
    c_ax.os = (L2 / 2) * zts;
    c_ax.src = in;
    c_ax.src.im = 1;                   /* odd sample = imaginary part */
    c_ax.dst = z;
    c_ax.dense = 0;
    c_ax.flags_in = c_ax.flags_out = 0;
    fa_emit_axis(p, &c_ax);
    if (p->failed || p->nsteps != n0 + 1) goto undo;

    nd = 0;
    d[nd].n = L1 / 2 + 1; d[nd].is = (L2 / 2) * zts; d[nd].os = 2; d[nd].tw = 1; d[nd].is_batch = 0; ++nd;
    for (j = 0; j < ax.nloops; ++j) {
        d[nd].n = ax.loops[j].n; d[nd].is = lts[j]; d[nd].os = ax.loops[j].os;
        d[nd].tw = 0; d[nd].is_batch = (j == ax.batch_loop); ++nd;
    }
    emit_pass(p, z, out, L2, zts, 2 * L1, d, nd, nl, FFTW_AMD_F_TW_IN | FFTW_AMD_F_REAL_DEC);
    if (p->failed) goto undo;
    s = &p->steps[p->nsteps - 1];
    if (s->variant != FFTW_AMD_K_R3 || s->tile != fa_hip_r3t_tile((int)L2) || s->tile_lo_n != 1 ||
        s->dim_n[0] != L1 / 2 + 1 || s->dim_tw[0] != 1 || s->batch_dim == 0)
        goto undo;
    buf_release(p, zbuf);
    return 1;
undo:
    if (p->failed) return 1;
    p->nsteps = n0;
    p->est_flops = flops0;
    buf_release(p, zbuf);
    return 0;
}

/* epi: 0 = store the half spectrum as complex numbers; FFTW_AMD_R2R_POST_* = the
   untangle step applies that r2r epilogue and writes reals of stride cs instead
   (even lengths only; returns 1 when the epilogue was fused, 0 when the caller
   still has to run it on the complex result)
   ps / pim (even lengths): the real input as pairs -- sample 2j at j*ps, sample 2j+1 at
   j*ps + pim; 0 / 0 = the plain strided array (ps = 2 rs, pim = rs).  Scratch inputs of the
   r2r emitters use ps = any, pim = 1: pairs stay adjacent even when other loops are
   innermost, so the register kernels take the first pass. */
static int emit_r2c_axis(plan *p, i64 nl, const fa_axis *axp, fa_loc in, i64 rs, fa_loc out, i64 cs, int epi,
                         i64 ps, i64 pim, int pre) {
    fa_axis ax = *axp;
    i64 half = nl / 2 + 1;
    int j;
    if (ps == 0) { ps = 2 * rs; pim = rs; }
    if (epi == 0 && pre == 0 && ps == 2 && pim == 1 && cs == 2 && emit_r2c_decimated(p, nl, &ax, in, out)) return 0;
    if (nl % 4 == 0 && nl >= 8 && ax.nloops < FA_MAXLOOPS && !getenv("FFTW_AMD_NO_RADIX4") &&
        ((axis_pass_count(p, nl / 4) < half_axis_pass_count(p, nl / 2) && radix4_passes_have_kernels(p, nl / 4)) ||
         getenv("FFTW_AMD_FORCE_RADIX4"))) {
        /* n = 4m: two complex DFTs of size m on (x[4j], x[4j+1]) and (x[4j+2], x[4j+3]),
           then the radix-4 untangle -- the reference's rdft2-ct-dit/4 + hc2cfdft_4 plan,
           chosen when m needs fewer passes than n/2 (n = 2^22: m = 2^20 is 1024 x 1024) */
        i64 m = nl / 4, zts, lts[FA_MAXLOOPS + 1], total;
        int zbuf, nd, vloop;
        fa_loc z;
        fa_axis q_ax = ax, lay;
        sdim d[FA_MAXLOOPS + 1];
        fftw_amd_step_desc *s;
        /* scratch [loops][v][m]: the vector index v rides as the innermost loop */
        vloop = q_ax.nloops++;
        q_ax.loops[vloop].n = 2;
        q_ax.loops[vloop].is = ps;
        q_ax.loops[vloop].os = 0;
        lay = q_ax;
        lay.is = 2;
        lay.loops[vloop].is = 1;           /* the pair index is innermost: Z[k][v], like the input */
        total = scratch_layout(&lay, m, &zts, lts);
        zbuf = buf_acquire(p, total);
        z.buf = zbuf; z.base = 0; z.im = 1;
        q_ax.dense = (pim == 1 && ps == 2);
        q_ax.n = m;
        q_ax.is = 2 * ps;
        q_ax.os = zts;
        q_ax.src = in;
        q_ax.src.im = pim;
        q_ax.dst = z;
        for (j = 0; j < q_ax.nloops; ++j) q_ax.loops[j].os = lts[j];
        fa_emit_axis(p, &q_ax);

        s = new_step(p, FFTW_AMD_STEP_R2C_POST4);
        s->src_buf = zbuf; s->src_base = 0; s->src_im = 1;
        s->dst_buf = out.buf; s->dst_base = out.base; s->dst_im = out.im;
        s->is_l = zts;
        s->os_l = cs;
        s->aux_n = nl;
        s->aux_valid = lts[vloop];
        r2r_fuse_tables(p, s, nl, epi);
        nd = 0;
        for (j = 0; j < ax.nloops; ++j) {
            d[nd].n = ax.loops[j].n; d[nd].is = lts[j]; d[nd].os = ax.loops[j].os;
            d[nd].tw = 0; d[nd].is_batch = (j == ax.batch_loop); ++nd;
        }
        s->kpos = order_dims_kpos(d, &nd, 1, cs);
        step_set_dims(p, s, d, nd, -1);
        p->est_flops += 20.0 * (double)m;
        buf_release(p, zbuf);
    } else if (nl % 2 == 0 && nl >= 2 && (pre != 0 || (ps == 2 && pim == 1)) &&
               (epi != 0 || (cs == 2 && out.im == 1)) &&
               (short_real_rows_tile(&ax, nl / 2, epi || pre, 1) > 0 || real_rows_tile(nl / 2, epi || pre) > 0) &&
               r2c_rows_layout_ok(p, &ax, in, out, epi, pre)) {
        /* contiguous real rows of a supported length: the half-length complex DFT and the
           untangle in ONE trip (r2crows.hpp) instead of a pass plus an untangle step */
        sdim d[FA_MAXLOOPS];
        int nd = 0;
        fftw_amd_step_desc *s;
        fa_loc src = in;
        src.im = 1;
        for (j = 0; j < ax.nloops; ++j) {
            d[nd].n = ax.loops[j].n; d[nd].is = ax.loops[j].is; d[nd].os = ax.loops[j].os;
            d[nd].tw = 0; d[nd].is_batch = (j == ax.batch_loop); ++nd;
        }
        emit_pass(p, src, out, nl / 2, pre ? rs : 2, epi ? cs : 2, d, nd, 0, FFTW_AMD_F_R2C_ROWS);
        s = &p->steps[p->nsteps - 1];
        s->variant = FFTW_AMD_K_R2C;
        s->aux_buf = pre ? pre : -1;     /* r2r pre-processing gathered inside the row (FFTW_AMD_R2R_PRE_*) */
        s->tile = short_real_rows_tile(&ax, nl / 2, epi || pre, 1) > 0 ? short_real_rows_tile(&ax, nl / 2, epi || pre, 1)
                                                                    : real_rows_tile(nl / 2, epi || pre);
        s->tile_lo_n = 1;
        /* aux_n = n, aux_valid = fused r2r epilogue (0: plain half spectrum), aux_base = index
           multiplier of the untangle twiddle in the table (4 with the modulus-4n table of the
           DCT-II / DST-II epilogues, see r2r_fuse_tables) */
        s->aux_n = nl;
        s->aux_valid = epi;
        if (epi == FFTW_AMD_R2R_POST_E10 || epi == FFTW_AMD_R2R_POST_O10) {
            tab_tw2(p, 4 * nl, &s->tw_lo, &s->tw_hi, &s->tw_shift);
            s->aux_base = 4;
        } else {
            tab_tw2(p, nl, &s->tw_lo, &s->tw_hi, &s->tw_shift);
            s->aux_base = 1;
        }
        p->est_flops += 8.0 * (double)(nl / 2);
        return epi != 0;
    } else if (nl % 2 == 0 && nl >= 2) {
        /* z[j] = x[2j] + i x[2j+1]; Z = DFT_{n/2}(z); untangle */
        i64 h = nl / 2, zts, lts[FA_MAXLOOPS], total;
        int zbuf, nd;
        fa_loc z;
        fa_axis half_ax = ax, lay;
        sdim d[FA_MAXLOOPS];
        fftw_amd_step_desc *s;
        lay = ax;
        lay.is = 1;
        total = scratch_layout(&lay, h, &zts, lts);
        zbuf = buf_acquire(p, total);
        z.buf = zbuf; z.base = 0; z.im = 1;

        half_ax.n = h;
        half_ax.is = ps;
        half_ax.os = zts;
        half_ax.src = in;
        half_ax.src.im = pim;    /* odd sample = imaginary part */
        half_ax.dst = z;
        for (j = 0; j < ax.nloops; ++j) half_ax.loops[j].os = lts[j];
        fa_emit_axis(p, &half_ax);

        s = new_step(p, FFTW_AMD_STEP_R2C_POST);
        s->src_buf = zbuf; s->src_base = 0; s->src_im = 1;
        s->dst_buf = out.buf; s->dst_base = out.base; s->dst_im = out.im;
        s->is_l = zts;
        s->os_l = cs;
        s->aux_n = nl;
        r2r_fuse_tables(p, s, nl, epi);
        nd = 0;
        for (j = 0; j < ax.nloops; ++j) {
            d[nd].n = ax.loops[j].n; d[nd].is = lts[j]; d[nd].os = ax.loops[j].os;
            d[nd].tw = 0; d[nd].is_batch = (j == ax.batch_loop); ++nd;
        }
        s->kpos = order_dims_kpos(d, &nd, 1, cs);
        step_set_dims(p, s, d, nd, -1);
        p->est_flops += 8.0 * (double)h;
        buf_release(p, zbuf);
    } else {
        /* odd length: full complex DFT of the real sequence, keep half */
        i64 fts, lts[FA_MAXLOOPS], total;
        int fbuf, nd;
        fa_loc f;
        fa_axis full_ax = ax, lay;
        sdim d[FA_MAXLOOPS];
        lay = ax;
        lay.is = 1;
        total = scratch_layout(&lay, nl, &fts, lts);
        fbuf = buf_acquire(p, total);
        f.buf = fbuf; f.base = 0; f.im = 1;
        full_ax.n = nl;
        full_ax.is = rs;
        full_ax.os = fts;
        full_ax.src = in;
        full_ax.dst = f;
        full_ax.flags_in = FFTW_AMD_F_REAL_IN;
        for (j = 0; j < ax.nloops; ++j) full_ax.loops[j].os = lts[j];
        fa_emit_axis(p, &full_ax);
        nd = 0;
        for (j = 0; j < ax.nloops; ++j) {
            d[nd].n = ax.loops[j].n; d[nd].is = lts[j]; d[nd].os = ax.loops[j].os;
            d[nd].tw = 0; d[nd].is_batch = (j == ax.batch_loop); ++nd;
        }
        emit_copy(p, FFTW_AMD_STEP_COPY, f, out, half, half, fts, cs, d, nd, 0, -1, -1);
        buf_release(p, fbuf);
        return 0;
    }
    return epi != 0;
}

static void build_r2c(plan *p) {
    int r = p->rank, a, j;
    i64 nl = p->dims[r - 1].n, half = nl / 2 + 1;
    i64 ext[FA_MAXRANK];
    fa_loc in = { 0, 0, 0 }, out = { 1, 0, p->out_im };
    fa_axis ax;
    for (j = 0; j < r; ++j) ext[j] = p->dims[j].n;
    ext[r - 1] = half;

    memset(&ax, 0, sizeof(ax));
    if (collect_loops(p, p->dims, r, r - 1, NULL, 0, &ax)) { p->failed = 1; return; }

    emit_r2c_axis(p, nl, &ax, in, p->dims[r - 1].is, out, p->dims[r - 1].os, 0, 0, 0, 0);

    for (a = r - 2; a >= 0; --a) {
        fa_axis cx;
        memset(&cx, 0, sizeof(cx));
        cx.n = p->dims[a].n;
        cx.is = cx.os = p->dims[a].os;
        cx.src = cx.dst = out;
        if (collect_loops(p, p->dims, r, a, ext, 1, &cx)) { p->failed = 1; return; }
        fa_emit_axis(p, &cx);
    }
}

/* c2r: complex backward DFTs over the leading dims (into scratch, so the
   caller's input survives), then half spectrum -> real along the last dim.
   p->dims[].is: strides of the complex array, .os: of the real array. */
/* half spectrum -> real along one axis (unnormalised backward).  ax: the loops
   (.is = strides in the complex source `cur`, .os = strides in the real
   destination); cs / rs: element strides of the transform index, in doubles. */
/* pro: 0 = `cur` holds the half spectrum as complex numbers; FFTW_AMD_R2R_PRE_* =
   `cur` is the user's real r2r input of stride cs and the tangle step applies
   that prologue while loading (even lengths only, see r2r_can_fuse)
   ps / pim: pair geometry of the real output, as in emit_r2c_axis */
static void emit_c2r_axis(plan *p, i64 nl, const fa_axis *axp, fa_loc cur, i64 cs, fa_loc out, i64 rs, int pro,
                          i64 ps, i64 pim, int post) {
    fa_axis ax = *axp;
    int j;
    if (ps == 0) { ps = 2 * rs; pim = rs; }
    if (nl % 4 == 0 && nl >= 8 && ax.nloops < FA_MAXLOOPS && !getenv("FFTW_AMD_NO_RADIX4") &&
        ((axis_pass_count(p, nl / 4) < half_axis_pass_count(p, nl / 2) && radix4_passes_have_kernels(p, nl / 4)) ||
         getenv("FFTW_AMD_FORCE_RADIX4"))) {
        /* transpose of the radix-4 r2c plan: tangle into two quarter-length
           spectra, two backward complex DFTs of size m straight into the real array */
        i64 m = nl / 4, zts, lts[FA_MAXLOOPS + 1], total;
        int zbuf, nd, vloop;
        fa_loc z;
        fa_axis q_ax = ax, lay;
        sdim d[FA_MAXLOOPS + 1];
        fftw_amd_step_desc *s;
        vloop = q_ax.nloops++;
        q_ax.loops[vloop].n = 2;
        q_ax.loops[vloop].is = 0;
        q_ax.loops[vloop].os = ps;
        lay = q_ax;
        lay.is = 2;
        lay.loops[vloop].is = 1;
        total = scratch_layout(&lay, m, &zts, lts);
        zbuf = buf_acquire(p, total);
        z.buf = zbuf; z.base = 0; z.im = 1;
        q_ax.dense = (pim == 1 && ps == 2);

        s = new_step(p, FFTW_AMD_STEP_C2R_PRE4);
        s->src_buf = cur.buf; s->src_base = cur.base; s->src_im = cur.im;
        s->dst_buf = zbuf; s->dst_base = 0; s->dst_im = 1;
        s->is_l = cs;
        s->os_l = zts;
        s->aux_n = nl;
        s->aux_valid = lts[vloop];
        r2r_fuse_tables(p, s, nl, pro);
        nd = 0;
        for (j = 0; j < ax.nloops; ++j) {
            d[nd].n = ax.loops[j].n; d[nd].is = ax.loops[j].is; d[nd].os = lts[j];
            d[nd].tw = 0; d[nd].is_batch = (j == ax.batch_loop); ++nd;
        }
        s->kpos = order_dims_kpos(d, &nd, 0, cs);
        step_set_dims(p, s, d, nd, -1);

        q_ax.n = m;
        q_ax.is = zts;
        q_ax.os = 2 * ps;
        q_ax.src = z;
        q_ax.dst = out;
        q_ax.dst.im = pim;
        q_ax.flags_in = FFTW_AMD_F_SWAP_IN;
        q_ax.flags_out = FFTW_AMD_F_SWAP_OUT;
        for (j = 0; j < q_ax.nloops; ++j) q_ax.loops[j].is = lts[j];
        fa_emit_axis(p, &q_ax);
        buf_release(p, zbuf);
    } else if (nl % 2 == 0 && nl >= 2 && (post != 0 || (ps == 2 && pim == 1)) &&
               (pro != 0 || (cs == 2 && cur.im == 1)) &&
               (short_real_rows_tile(&ax, nl / 2, pro || post, 0) > 0 || real_rows_tile(nl / 2, pro || post) > 0) &&
               c2r_rows_layout_ok(p, &ax, cur, out, pro, post)) {
        /* contiguous rows of a supported length: tangle + backward half-length DFT in ONE trip */
        sdim d[FA_MAXLOOPS];
        int nd = 0;
        fftw_amd_step_desc *s;
        fa_loc dstp = out;
        dstp.im = 1;
        for (j = 0; j < ax.nloops; ++j) {
            d[nd].n = ax.loops[j].n; d[nd].is = ax.loops[j].is; d[nd].os = ax.loops[j].os;
            d[nd].tw = 0; d[nd].is_batch = (j == ax.batch_loop); ++nd;
        }
        emit_pass(p, cur, dstp, nl / 2, pro ? cs : 2, post ? rs : 2, d, nd, 0, FFTW_AMD_F_C2R_ROWS);
        s = &p->steps[p->nsteps - 1];
        s->variant = FFTW_AMD_K_C2R;
        s->aux_buf = post ? post : -1;   /* r2r output shuffle done by the kernel's store (FFTW_AMD_R2R_POST_E01 / O01) */
        s->tile = short_real_rows_tile(&ax, nl / 2, pro || post, 0) > 0 ? short_real_rows_tile(&ax, nl / 2, pro || post, 0)
                                                                        : real_rows_tile(nl / 2, pro || post);
        s->tile_lo_n = 1;
        s->aux_n = nl;              /* as in the r2c rows step: n, fused r2r prologue, twiddle multiplier */
        s->aux_valid = pro;
        if (pro == FFTW_AMD_R2R_PRE_E01 || pro == FFTW_AMD_R2R_PRE_O01) {
            tab_tw2(p, 4 * nl, &s->tw_lo, &s->tw_hi, &s->tw_shift);
            s->aux_base = 4;
        } else {
            tab_tw2(p, nl, &s->tw_lo, &s->tw_hi, &s->tw_shift);
            s->aux_base = 1;
        }
        p->est_flops += 8.0 * (double)(nl / 2);
    } else if (nl % 2 == 0 && nl >= 2) {
        i64 h = nl / 2, zts, lts[FA_MAXLOOPS], total;
        int zbuf, nd;
        fa_loc z;
        fa_axis half_ax, lay;
        sdim d[FA_MAXLOOPS];
        fftw_amd_step_desc *s;
        lay = ax;
        lay.is = 1;
        total = scratch_layout(&lay, h, &zts, lts);
        zbuf = buf_acquire(p, total);
        z.buf = zbuf; z.base = 0; z.im = 1;

        s = new_step(p, FFTW_AMD_STEP_C2R_PRE);
        s->src_buf = cur.buf; s->src_base = cur.base; s->src_im = cur.im;
        s->dst_buf = zbuf; s->dst_base = 0; s->dst_im = 1;
        s->is_l = cs;
        s->os_l = zts;
        s->aux_n = nl;
        r2r_fuse_tables(p, s, nl, pro);
        nd = 0;
        for (j = 0; j < ax.nloops; ++j) {
            d[nd].n = ax.loops[j].n; d[nd].is = ax.loops[j].is; d[nd].os = lts[j];
            d[nd].tw = 0; d[nd].is_batch = (j == ax.batch_loop); ++nd;
        }
        s->kpos = order_dims_kpos(d, &nd, 0, cs);
        step_set_dims(p, s, d, nd, -1);

        half_ax = ax;
        half_ax.n = h;
        half_ax.is = zts;
        half_ax.os = ps;
        half_ax.src = z;
        half_ax.dst = out;
        half_ax.dst.im = pim;
        half_ax.flags_in = FFTW_AMD_F_SWAP_IN;
        half_ax.flags_out = FFTW_AMD_F_SWAP_OUT;
        for (j = 0; j < ax.nloops; ++j) half_ax.loops[j].is = lts[j];
        fa_emit_axis(p, &half_ax);
        buf_release(p, zbuf);
    } else {
        i64 fts, lts[FA_MAXLOOPS], total;
        int fbuf, nd;
        fa_loc f;
        fa_axis full_ax, lay;
        sdim d[FA_MAXLOOPS];
        lay = ax;
        lay.is = 1;
        total = scratch_layout(&lay, nl, &fts, lts);
        fbuf = buf_acquire(p, total);
        f.buf = fbuf; f.base = 0; f.im = 1;
        nd = 0;
        for (j = 0; j < ax.nloops; ++j) {
            d[nd].n = ax.loops[j].n; d[nd].is = ax.loops[j].is; d[nd].os = lts[j];
            d[nd].tw = 0; d[nd].is_batch = (j == ax.batch_loop); ++nd;
        }
        emit_copy(p, FFTW_AMD_STEP_HERM_EXPAND, cur, f, nl, nl, cs, fts, d, nd, 0, -1, -1);
        full_ax = ax;
        full_ax.n = nl;
        full_ax.is = fts;
        full_ax.os = rs;
        full_ax.src = f;
        full_ax.dst = out;
        full_ax.flags_in = FFTW_AMD_F_SWAP_IN;
        /* backward by the swap identity: the real result is the imaginary
           slot of the swapped output, so swap on store and keep the real part */
        full_ax.flags_out = FFTW_AMD_F_SWAP_OUT | FFTW_AMD_F_REAL_OUT;
        for (j = 0; j < ax.nloops; ++j) full_ax.loops[j].is = lts[j];
        fa_emit_axis(p, &full_ax);
        buf_release(p, fbuf);
    }
}

static void build_c2r(plan *p) {
    int r = p->rank, a, j;
    i64 nl = p->dims[r - 1].n, half = nl / 2 + 1;
    i64 ext[FA_MAXRANK];
    fa_loc in = { 0, 0, p->in_im }, out = { 1, 0, 0 };
    fa_loc cur = in;
    fa_dim cdims[FA_MAXRANK];     /* current layout of the half-spectrum array */
    fa_dim chd[FA_MAXRANK];
    int cbuf = -1;
    fa_axis ax;

    for (j = 0; j < r; ++j) { ext[j] = p->dims[j].n; cdims[j] = p->dims[j]; cdims[j].os = p->dims[j].is; }
    ext[r - 1] = half;
    for (j = 0; j < p->hrank; ++j) { chd[j] = p->hdims[j]; chd[j].os = p->hdims[j].is; }

    if (r > 1) {
        /* dense scratch for the half spectrum: [batch][d0]...[half] */
        i64 stride = 2, total;
        fa_dim sd[FA_MAXRANK], sh[FA_MAXRANK];
        plan view;
        int first = 1;
        for (j = r - 1; j >= 0; --j) { sd[j].n = ext[j]; sd[j].os = stride; stride *= ext[j]; }
        for (j = p->hrank - 1; j >= 1; --j) { sh[j].n = p->hdims[j].n; sh[j].os = stride; stride *= p->hdims[j].n; }
        if (p->hrank) { sh[0].n = p->hdims[0].n; sh[0].os = stride; stride *= p->chunk; }
        total = stride;
        cbuf = buf_acquire(p, total);
        /* leading dims: in -> scratch for the first, in place afterwards */
        for (a = r - 2; a >= 0; --a) {
            fa_axis cx;
            fa_dim td[FA_MAXRANK];
            memset(&cx, 0, sizeof(cx));
            for (j = 0; j < r; ++j) {
                td[j].n = ext[j];
                td[j].is = first ? p->dims[j].is : sd[j].os;
                td[j].os = sd[j].os;
            }
            view = *p;
            for (j = 0; j < p->hrank; ++j) {
                view.hdims[j].is = first ? p->hdims[j].is : sh[j].os;
                view.hdims[j].os = sh[j].os;
            }
            cx.n = p->dims[a].n;
            cx.is = td[a].is;
            cx.os = td[a].os;
            cx.src = first ? in : (fa_loc){ cbuf, 0, 1 };
            cx.dst = (fa_loc){ cbuf, 0, 1 };
            cx.flags_in = FFTW_AMD_F_SWAP_IN;
            cx.flags_out = FFTW_AMD_F_SWAP_OUT;
            if (collect_loops(&view, td, r, a, ext, 0, &cx)) { p->failed = 1; return; }
            fa_emit_axis(p, &cx);
            first = 0;
        }
        cur = (fa_loc){ cbuf, 0, 1 };
        for (j = 0; j < r; ++j) { cdims[j].is = sd[j].os; }
        for (j = 0; j < p->hrank; ++j) chd[j].is = sh[j].os;
    }

    /* loops of the last-dim c2r: source strides = current complex layout,
       destination strides = the real output array */
    memset(&ax, 0, sizeof(ax));
    {
        plan view = *p;
        fa_dim td[FA_MAXRANK];
        for (j = 0; j < r; ++j) { td[j].n = p->dims[j].n; td[j].is = cdims[j].is; td[j].os = p->dims[j].os; }
        for (j = 0; j < p->hrank; ++j) { view.hdims[j].is = chd[j].is; view.hdims[j].os = p->hdims[j].os; }
        if (collect_loops(&view, td, r, r - 1, NULL, 0, &ax)) { p->failed = 1; return; }
    }

    emit_c2r_axis(p, nl, &ax, cur, cdims[r - 1].is, out, p->dims[r - 1].os, 0, 0, 0, 0);
    if (cbuf >= 0) buf_release(p, cbuf);
}


/* ------------------------------------------------------------------ r2r */
/* One r2r axis = PRE step (user reals -> inner input), an inner real or complex
   DFT on scratch, POST step (inner output -> user reals).  The reference gets the
   same transforms from R2HC/HC2R children with pre/post loops around them:
   reodft010e-r2hc (fftw/fftw_api.c:12465-12660), redft00e-r2hc-pad (:11932-11965),
   rodft00e-r2hc-pad (:14012-14050), reodft11e-radix2 (:13362-13640), reodft11e-r2hc-odd
   (:13097-13230), dht-r2hc (:6365-6387).  Derivations of the index maps: DESIGN.md
   section 9. */

static fftw_amd_step_desc *emit_r2r_step(plan *p, int mode, i64 n, i64 K, i64 twmod,
                                         fa_loc src, i64 is_k, fa_loc dst, i64 os_k,
                                         const fa_axis *ax, const i64 *lis, const i64 *los,
                                         int order_by_dst) {
    fftw_amd_step_desc *s = new_step(p, FFTW_AMD_STEP_R2R);
    sdim d[FA_MAXLOOPS + 1];
    int cnt = 0, i;
    s->variant = mode;
    s->src_buf = src.buf; s->src_base = src.base; s->src_im = src.im;
    s->dst_buf = dst.buf; s->dst_base = dst.base; s->dst_im = dst.im;
    s->is_l = is_k;
    s->os_l = os_k;
    s->aux_n = n;
    s->aux_valid = K;
    if (twmod) tab_tw2(p, twmod, &s->tw_lo, &s->tw_hi, &s->tw_shift);
    /* flatten order: smallest user-side stride fastest, the batch loop last */
    for (i = 0; i < ax->nloops; ++i) {
        d[cnt].n = ax->loops[i].n;
        d[cnt].is = lis[i];
        d[cnt].os = los[i];
        d[cnt].tw = 0;
        d[cnt].is_batch = (i == ax->batch_loop);
        ++cnt;
    }
    s->kpos = order_dims_kpos(d, &cnt, order_by_dst, order_by_dst ? os_k : is_k);
    step_set_dims(p, s, d, cnt, -1);
    p->est_flops += 4.0 * (double)K;
    return s;
}

/* ax: loops with .is = user source strides, .os = user destination strides */
static void emit_r2r_axis(plan *p, int kind, i64 n, const fa_axis *axp, fa_loc in, i64 rs,
                          fa_loc out, i64 os) {
    enum { IN_R2C, IN_C2R, IN_C2C };
    int pre = 0, post = 0, inner = IN_R2C, j, nl = axp->nloops, fuse_pre = 0, fuse_post = 0, rows_pre = 0, rows_post = 0;
    i64 N = n, cntA = 0, unitA = 1, cntB = 0, unitB = 1, twmod = 0, Kpre = 0, Kpost = 0;
    i64 lis_user[FA_MAXLOOPS], los_user[FA_MAXLOOPS], ltsA[FA_MAXLOOPS], ltsB[FA_MAXLOOPS];
    i64 tsA = 0, tsB = 0, psA = 0, pimA = 0, psB = 0, pimB = 0;
    int abuf = -1, bbuf = -1, pairA = 0, pairB = 0;
    fa_loc A = { -1, 0, 1 }, B = { -1, 0, 1 };
    fa_axis lay, iax;
    for (j = 0; j < nl; ++j) { lis_user[j] = axp->loops[j].is; los_user[j] = axp->loops[j].os; }

    switch (kind) {
    case FFTW_R2HC: post = FFTW_AMD_R2R_POST_R2HC; goto via_r2c;
    case FFTW_DHT:  post = FFTW_AMD_R2R_POST_DHT;
    via_r2c:
        inner = IN_R2C; N = n; cntB = n / 2 + 1; unitB = 2; Kpost = n / 2 + 1;
        break;
    case FFTW_HC2R:
        pre = FFTW_AMD_R2R_PRE_HC2R; inner = IN_C2R; N = n; cntA = n / 2 + 1; unitA = 2; Kpre = n / 2 + 1;
        break;
    case FFTW_REDFT10: pre = FFTW_AMD_R2R_PRE_E10; post = FFTW_AMD_R2R_POST_E10; goto makhoul_f;
    case FFTW_RODFT10: pre = FFTW_AMD_R2R_PRE_O10; post = FFTW_AMD_R2R_POST_O10;
    makhoul_f:
        inner = IN_R2C; N = n; cntA = n; unitA = 1; cntB = n / 2 + 1; unitB = 2;
        Kpre = (n + 1) / 2; Kpost = n / 2 + 1; twmod = 4 * n;
        break;
    case FFTW_REDFT01: pre = FFTW_AMD_R2R_PRE_E01; post = FFTW_AMD_R2R_POST_E01; goto makhoul_b;
    case FFTW_RODFT01: pre = FFTW_AMD_R2R_PRE_O01; post = FFTW_AMD_R2R_POST_O01;
    makhoul_b:
        inner = IN_C2R; N = n; cntA = n / 2 + 1; unitA = 2; cntB = n; unitB = 1;
        Kpre = n / 2 + 1; Kpost = (n + 1) / 2; twmod = 4 * n;
        break;
    case FFTW_REDFT00:
        if (n < 2) { p->failed = 1; return; }
        pre = FFTW_AMD_R2R_PRE_E00; post = FFTW_AMD_R2R_POST_E00;
        inner = IN_R2C; N = 2 * (n - 1); cntA = N; unitA = 1; cntB = N / 2 + 1; unitB = 2;
        Kpre = N; Kpost = n;
        break;
    case FFTW_RODFT00:
        pre = FFTW_AMD_R2R_PRE_O00; post = FFTW_AMD_R2R_POST_O00;
        inner = IN_R2C; N = 2 * (n + 1); cntA = N; unitA = 1; cntB = N / 2 + 1; unitB = 2;
        Kpre = N; Kpost = n;
        break;
    case FFTW_REDFT11:
    case FFTW_RODFT11: {
        int odd = (kind == FFTW_RODFT11);
        inner = IN_C2C; twmod = 8 * n; unitA = 2;
        if (n % 2 == 0) {
            pre = odd ? FFTW_AMD_R2R_PRE_O11 : FFTW_AMD_R2R_PRE_E11;
            post = odd ? FFTW_AMD_R2R_POST_O11 : FFTW_AMD_R2R_POST_E11;
            N = n / 2; cntA = N; Kpre = N; Kpost = N;
        } else {
            pre = odd ? FFTW_AMD_R2R_PRE_O11ODD : FFTW_AMD_R2R_PRE_E11ODD;
            post = odd ? FFTW_AMD_R2R_POST_O11ODD : FFTW_AMD_R2R_POST_E11ODD;
            N = 2 * n; cntA = N; Kpre = N; Kpost = n;
        }
        break;
    }
    default:
        p->failed = 1;
        return;
    }

    /* even inner length: the untangle / tangle step of the real transform does the
       r2r post / pre processing itself, one pass over the data less */
    if (inner == IN_R2C && post && r2r_can_fuse(N)) { fuse_post = 1; cntB = 0; }
    /* short rows: the fused real-rows kernel also gathers the pre-processed sequence from the
       user's row itself -- the whole r2r axis is one trip */
    if (fuse_post && pre && pre != FFTW_AMD_R2R_PRE_HC2R && fa_hip_r2c_rows_tile((int)(N / 2)) > 0) {
        fa_axis tax = *axp;
        if (r2c_rows_layout_ok(p, &tax, in, out, post, pre)) { rows_pre = pre; cntA = 0; }
    }
    if (inner == IN_C2R && pre && r2r_can_fuse(N)) { fuse_pre = 1; cntA = 0; }
    /* short rows: the fused c2r rows kernel also does the DCT-III / DST-III output shuffle */
    if (fuse_pre && post && fa_hip_r2c_rows_tile((int)(N / 2)) > 0) {
        fa_axis tax = *axp;
        if (c2r_rows_layout_ok(p, &tax, in, out, pre, post)) { rows_post = post; cntB = 0; }
    }

    /* Real scratch sequences of even length are laid out as adjacent pairs
       (sample 2j, 2j+1) = one interleaved complex number of stride ts: whatever loops end
       up innermost, the inner real transform then reads / writes 16-byte elements and its
       first / last pass can be a register kernel.  pairA / pairB = 1 marks that layout;
       odd lengths keep a plain real stride (pair stride 2 ts, pair im ts). */
    if (cntA) {
        i64 total;
        lay = *axp;
        lay.is = rs;
        if (unitA == 1 && cntA % 2 == 0) {
            pairA = 1;
            total = scratch_layout_u(&lay, cntA / 2, 2, &tsA, ltsA);
        } else {
            total = scratch_layout_u(&lay, cntA, unitA, &tsA, ltsA);
        }
        abuf = buf_acquire(p, total);
        A.buf = abuf;
    }
    if (cntB) {
        i64 total;
        lay = *axp;
        lay.is = os;
        for (j = 0; j < nl; ++j) lay.loops[j].is = los_user[j];
        if (unitB == 1 && cntB % 2 == 0) {
            pairB = 1;
            total = scratch_layout_u(&lay, cntB / 2, 2, &tsB, ltsB);
        } else {
            total = scratch_layout_u(&lay, cntB, unitB, &tsB, ltsB);
        }
        bbuf = buf_acquire(p, total);
        B.buf = bbuf;
    }
    /* pair geometry as the steps see it: (pair stride, distance inside the pair) */
    psA = (unitA == 1) ? (pairA ? tsA : 2 * tsA) : 0;  pimA = (unitA == 1) ? (pairA ? 1 : tsA) : 0;
    psB = (unitB == 1) ? (pairB ? tsB : 2 * tsB) : 0;  pimB = (unitB == 1) ? (pairB ? 1 : tsB) : 0;

    if (pre && !fuse_pre && !rows_pre) {
        fa_loc dA = A;
        if (unitA == 1) dA.im = pimA;          /* real sequence: element j at (j >> 1) psA + (j & 1) pimA */
        emit_r2r_step(p, pre, n, Kpre, twmod, in, rs, dA, unitA == 1 ? psA : tsA, axp, lis_user, ltsA, 0);
    }

    iax = *axp;
    iax.flags_in = iax.flags_out = 0;
    iax.dense = 0;
    if (inner == IN_R2C) {
        fa_loc s = pre ? A : in;
        for (j = 0; j < nl; ++j) {
            iax.loops[j].is = (pre && !rows_pre) ? ltsA[j] : lis_user[j];
            iax.loops[j].os = fuse_post ? los_user[j] : ltsB[j];
        }
        s.im = 0;
        if (rows_pre) {
            emit_r2c_axis(p, N, &iax, in, rs, out, os, post, 0, 0, rows_pre);
            post = 0;
        } else if (fuse_post) {
            emit_r2c_axis(p, N, &iax, s, pre ? pimA : rs, out, os, post, pre ? psA : 0, pre ? pimA : 0, 0);
            post = 0;
        } else {
            emit_r2c_axis(p, N, &iax, s, pre ? pimA : rs, B, tsB, 0, pre ? psA : 0, pre ? pimA : 0, 0);
        }
    } else if (inner == IN_C2R) {
        fa_loc d = post ? B : out;
        for (j = 0; j < nl; ++j) {
            iax.loops[j].is = fuse_pre ? lis_user[j] : ltsA[j];
            iax.loops[j].os = post ? ltsB[j] : los_user[j];
        }
        d.im = 0;
        if (rows_post) {
            for (j = 0; j < nl; ++j) iax.loops[j].os = los_user[j];
            emit_c2r_axis(p, N, &iax, in, rs, out, os, pre, 0, 0, rows_post);
            post = 0;
        } else if (fuse_pre) emit_c2r_axis(p, N, &iax, in, rs, d, post ? pimB : os, pre, post ? psB : 0, post ? pimB : 0, 0);
        else emit_c2r_axis(p, N, &iax, A, tsA, d, post ? pimB : os, 0, post ? psB : 0, post ? pimB : 0, 0);
    } else {
        for (j = 0; j < nl; ++j) { iax.loops[j].is = ltsA[j]; iax.loops[j].os = ltsA[j]; }
        iax.n = N; iax.is = tsA; iax.os = tsA;
        iax.src = A; iax.dst = A;
        fa_emit_axis(p, &iax);
        B = A; tsB = tsA; unitB = 2;
        for (j = 0; j < nl; ++j) ltsB[j] = ltsA[j];
    }

    if (post) {
        fa_loc sB = B;
        if (unitB == 1) sB.im = pimB;          /* real sequence read as pairs, see above */
        emit_r2r_step(p, post, n, Kpost, twmod, sB, unitB == 1 ? psB : tsB, out, os, axp, ltsB, los_user, 1);
    }
    if (abuf >= 0) buf_release(p, abuf);
    if (bbuf >= 0) buf_release(p, bbuf);
}

/* separable r2r: every dim gets its own kind (reference problem_rdft with
   kind[] per dim, fftw/fftw_api.c:9100-9150); the first processed axis moves the
   data from the input to the output array, the others work in place there */
static void build_r2r(plan *p) {
    int a, first = 1;
    fa_loc in = { 0, 0, 0 }, out = { 1, 0, 0 };
    if (p->rank == 0) {
        /* rank 0: strided copy of reals (reference rdft rank0 solver, fftw/fftw_api.c:9655-9720) */
        fa_axis ax;
        sdim d[FA_MAXLOOPS];
        int nd;
        memset(&ax, 0, sizeof(ax));
        if (collect_loops(p, p->dims, 0, -1, NULL, 0, &ax)) { p->failed = 1; return; }
        nd = loops_to_sdims(&ax, d, 0);
        emit_copy(p, FFTW_AMD_STEP_COPY, in, out, 1, 1, 0, 0, d, nd,
                  FFTW_AMD_F_REAL_IN | FFTW_AMD_F_REAL_OUT, -1, -1);
        return;
    }
    for (a = p->rank - 1; a >= 0; --a) {
        fa_axis ax;
        memset(&ax, 0, sizeof(ax));
        if (collect_loops(p, p->dims, p->rank, a, NULL, !first, &ax)) { p->failed = 1; return; }
        emit_r2r_axis(p, p->kinds[a], p->dims[a].n, &ax, first ? in : out,
                      first ? p->dims[a].is : p->dims[a].os, out, p->dims[a].os);
        if (p->failed) return;
        first = 0;
    }
}

static void build_steps(plan *p) {
    switch (p->type) {
    case FA_C2C: build_c2c(p); break;
    case FA_R2C: build_r2c(p); break;
    case FA_C2R: build_c2r(p); break;
    case FA_R2R: build_r2r(p); break;
    default: p->failed = 1;
    }
}

/* Cache policy of the streams that touch the caller's arrays (FFTW_AMD_F_NT_IN / NT_OUT).
   An array that one execution reads once (or writes and never reads back) is moved with
   nontemporal accesses when the batch is too large to stay cached anyway: the Infinity
   Cache (256 MiB) then keeps the scratch image between two passes of a chunk instead of
   input / output lines nobody asks for again (tests/micro/membw3.hip).  Small problems keep
   plain accesses so that a consumer kernel still finds the output on chip. */
static int g_nt_policy = -1;      /* FFTW_AMD_NT: 0 never, 1 by size (default), 2 always */
static void mark_streaming_accesses(plan *p) {
    int i, j;
    double touched = ((double)(p->in_hi - p->in_lo) + (double)(p->out_hi - p->out_lo)) * sizeof(double);
    if (g_nt_policy < 0) { const char *e = getenv("FFTW_AMD_NT"); g_nt_policy = e ? atoi(e) : 1; }
    if (g_nt_policy == 0 || (g_nt_policy == 1 && touched < 384.0 * 1048576.0)) return;
    for (i = 0; i < p->nsteps; ++i) {
        fftw_amd_step_desc *s = &p->steps[i];
        if (s->src_buf == 0) {
            int readers = 0;
            for (j = 0; j < p->nsteps; ++j) readers += (p->steps[j].src_buf == 0) || (p->inplace && p->steps[j].src_buf == 1);
            if (readers == 1) s->flags |= FFTW_AMD_F_NT_IN;
        }
        if (s->dst_buf == 1) {
            int later = 0;
            for (j = i + 1; j < p->nsteps; ++j)
                later += p->steps[j].src_buf == 1 || p->steps[j].aux_buf == 1 || (p->inplace && p->steps[j].src_buf == 0);
            if (!later) s->flags |= FFTW_AMD_F_NT_OUT;
        }
    }
}

static int fa_build_steps_sized(plan *p);

int fa_build(plan *p) {
    int r = fa_build_steps_sized(p);
    if (r == 0 && p->batch > 0) mark_streaming_accesses(p);
    return r;
}

static int fa_build_steps_sized(plan *p) {
    i64 per_elem = 0;
    int i;
    p->batch = p->hrank ? p->hdims[0].n : 1;
    if (p->batch == 0) {   /* howmany == 0: a valid plan that does nothing */
        plan_reset_build(p);
        p->chunk = 0;
        return 0;
    }
    /* dry run at one batch element to size the scratch, then pick the chunk */
    plan_reset_build(p);
    p->chunk = 1;
    build_steps(p);
    if (p->failed) return -1;
    for (i = 2; i < p->nbufs; ++i) per_elem += p->buf_reals[i];
    if (per_elem == 0 || p->batch == 1 || p->single_chunk) {
        p->chunk = p->batch;
    } else {
        i64 c = (i64)(p->cfg.chunk_bytes / ((size_t)per_elem * sizeof(double)));
        if (c < 1) c = 1;
        if (c > p->batch) c = p->batch;
        p->chunk = c;
    }
    if (p->chunk != 1) {
        plan_reset_build(p);
        build_steps(p);
        if (p->failed) return -1;
        /* a plan can turn out scratch-free only once the batch loop has more than one entry
           (the fused rows kernels need rows to tile over): then nothing limits the chunk */
        per_elem = 0;
        for (i = 2; i < p->nbufs; ++i) per_elem += p->buf_reals[i];
        if (per_elem == 0 && p->chunk != p->batch) {
            p->chunk = p->batch;
            plan_reset_build(p);
            build_steps(p);
            if (p->failed) return -1;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- device */

static plan *make_c2c_1d_contig(i64 n) {
    plan *q = fa_plan_new();
    q->type = FA_C2C;
    q->sign = FFTW_FORWARD;
    q->rank = 1;
    q->dims[0].n = n; q->dims[0].is = 2; q->dims[0].os = 2;
    q->hrank = 0;
    if (fa_build(q)) { fa_plan_free(q); return NULL; }
    return q;
}

int fa_device_init(plan *p) {
    int i;
    if (p->dev_ready) return 0;
    if (fa_hip_device_count() <= 0) {
        fprintf(stderr, "fftw3_amd: no HIP device available: this executor has no CPU "
                        "fallback; fftw_execute cannot run\n");
        return -1;
    }
    for (i = 0; i < p->ntabs; ++i) {
        fa_table *t = &p->tabs[i];
        size_t bytes = (size_t)t->len * (t->kind == FA_TAB_PERM ? sizeof(i64) : 2 * sizeof(double));
        if (t->dev) continue;
        t->dev = fa_hip_malloc(bytes);
        if (!t->dev) return -1;
        if (t->kind == FA_TAB_DFT_OF) continue;    /* second sweep */
        fa_hip_memcpy_h2d(t->dev, t->host, bytes, p->stream);
    }
    fa_hip_stream_sync(p->stream);
    for (i = 0; i < p->ntabs; ++i) {
        fa_table *t = &p->tabs[i];
        if (t->kind != FA_TAB_DFT_OF) continue;
        {
            /* the convolution kernel is itself a DFT: compute it with a
               nested plan on the device, keep a host copy for introspection */
            plan *q = make_c2c_1d_contig(t->n);
            size_t bytes = (size_t)t->len * 2 * sizeof(double);
            if (!q) return -1;
            q->stream = p->stream;
            fa_run(q, (double *)p->tabs[t->src].dev, (double *)p->tabs[t->src].dev + 1,
                   (double *)t->dev, (double *)t->dev + 1);
            if (!t->host) t->host = malloc(bytes);
            fa_hip_memcpy_d2h(t->host, t->dev, bytes, p->stream);
            fa_hip_stream_sync(p->stream);
            fa_plan_free(q);
        }
    }
    /* chunk pipeline: worth it when there are several chunks of >= 2 steps */
    p->nslots = 1;
    p->pair = 0;
    {
        /* batched 1024 x 1024: pass 2 of chunk c-1 and pass 1 of chunk c share a launch (two scratch slots) */
        static int pair_mode = -1;     /* FFTW_AMD_PAIR=0 switches it off */
        if (pair_mode < 0) { const char *e = getenv("FFTW_AMD_PAIR"); pair_mode = e ? atoi(e) : 1; }
        if (pair_mode && !p->cfg.pipeline && p->cfg.lanes <= 1 && p->nsteps == 2 && p->nbufs == 3 && p->chunk > 0 && p->chunk < p->batch &&
            !p->single_chunk &&
            p->steps[0].kind == FFTW_AMD_STEP_PASS && p->steps[1].kind == FFTW_AMD_STEP_PASS &&
            p->steps[0].variant == FFTW_AMD_K_P1024 && p->steps[1].variant == FFTW_AMD_K_P1024 &&
            p->steps[0].src_buf == 0 && p->steps[0].dst_buf == 2 && p->steps[1].src_buf == 2 && p->steps[1].dst_buf == 1 &&
            p->steps[0].tw_n == 0 && p->steps[1].tw_n != 0 && (p->steps[1].flags & FFTW_AMD_F_TW_IN) &&
            p->steps[0].tile_lo_n <= 1 && p->steps[1].tile_lo_n <= 1 &&
            p->steps[0].batch_dim >= 0 && p->steps[1].batch_dim >= 0) {
            p->pair = 1;
            p->nslots = 2;
        }
    }
    if (!p->pair && p->chunk > 0 && p->nsteps >= 2 && (p->batch + p->chunk - 1) / p->chunk >= 3 &&
        !p->single_chunk && p->cfg.pipeline) {
        i64 per = 0;
        for (i = 2; i < p->nbufs; ++i) per += p->buf_reals[i];
        if (per > 0 && (size_t)per * sizeof(double) * 3 <= ((size_t)8 << 30)) {
            p->nslots = 3;
            p->split = p->nsteps / 2;
            p->pstream[0] = fa_hip_stream_create();
            p->pstream[1] = fa_hip_stream_create();
            for (i = 0; i < 4; ++i) { p->ev_a[i] = fa_hip_event_create(); p->ev_b[i] = fa_hip_event_create(); }
            p->ev_begin = fa_hip_event_create();
            p->ev_end[0] = fa_hip_event_create();
            p->ev_end[1] = fa_hip_event_create();
        }
    }
    /* chunk lanes: chunk c runs all its steps, in order, on stream c % lanes in scratch slot c % lanes; the
       lanes share the machine, so the tail of one lane's launch overlaps the other lane's next launch and
       no dependency crosses the lanes */
    p->lanes = 1;
    if (!p->pair && p->nslots == 1 && p->cfg.lanes > 1 && p->chunk > 0 && p->nsteps >= 2 && p->nbufs > 2 &&
        !p->single_chunk && (p->batch + p->chunk - 1) / p->chunk >= 2) {
        i64 nch = (p->batch + p->chunk - 1) / p->chunk, per = 0;
        for (i = 2; i < p->nbufs; ++i) per += p->buf_reals[i];
        /* a chunk whose scratch is beyond the budget anyway (one transform with hundreds of MB of scratch: two of
           them in flight leave nothing on the die) gains nothing from a second lane and loses to the doubled
           footprint -- cfg4, n = 15 375 360, 246 MB of scratch per transform: 67.0 ms serial, 71 ... 76 ms with two
           lanes; one 4096 x 4096 image (256 MiB): 14.7 serial, 14.5 with two (tools/perf/perf_lanes_gate.sh) */
        if ((size_t)per * sizeof(double) > p->cfg.chunk_bytes) nch = 0;
        p->lanes = p->cfg.lanes > FA_MAXLANES ? FA_MAXLANES : p->cfg.lanes;
        if (p->lanes > nch) p->lanes = nch > 0 ? (int)nch : 1;
        p->nslots = p->lanes;
        for (i = 0; p->lanes > 1 && i < p->lanes; ++i) { p->pstream[i] = fa_hip_stream_create(); p->ev_end[i] = fa_hip_event_create(); }
        if (p->lanes > 1) p->ev_begin = fa_hip_event_create();
    }
    for (i = 2; i < p->nbufs; ++i)
        if (!p->dbuf[i]) {
            p->dbuf[i] = (double *)fa_hip_malloc((size_t)p->buf_reals[i] * sizeof(double) * (size_t)p->nslots);
            if (!p->dbuf[i]) return -1;
        }
    p->dev_ready = 1;
    return 0;
}

/* How a host array is laid out in its device staging buffer.  Offsets count doubles relative to
   the pointer of the "real" part; im is the caller's real -> imaginary distance. */
typedef struct {
    int two;        /* 1: two packed planes of `span` doubles each, 0: one span [lo, hi] covering both parts */
    i64 lo, hi;     /* touched offsets (two: of each plane; one span: of both parts together) */
    i64 span;
    i64 im_dev;     /* real -> imaginary distance inside the staging buffer */
} stage_layout;

static stage_layout stage_layout_of(i64 lo, i64 hi, i64 im) {
    stage_layout L;
    L.span = hi - lo + 1;
    L.two = im != 0 && iabs(im) >= L.span;
    if (L.two) {
        L.lo = lo; L.hi = hi;
        L.im_dev = im > 0 ? L.span : -L.span;
    } else {
        L.lo = im < 0 ? lo + im : lo;
        L.hi = im > 0 ? hi + im : hi;
        L.span = L.hi - L.lo + 1;
        L.im_dev = im;
    }
    return L;
}
static size_t stage_bytes(const stage_layout *L) { return (size_t)L->span * (L->two ? 2 : 1) * sizeof(double); }
/* device address that plays the role of the caller's real-part pointer */
static double *stage_origin(const stage_layout *L, double *st) {
    return st - L->lo + ((L->two && L->im_dev < 0) ? L->span : 0);
}
static void stage_h2d(const stage_layout *L, double *st, const double *re, i64 im, void *stream) {
    double *o = stage_origin(L, st);
    if (L->two) {
        fa_hip_memcpy_h2d(o + L->lo, re + L->lo, (size_t)L->span * sizeof(double), stream);
        fa_hip_memcpy_h2d(o + L->im_dev + L->lo, re + im + L->lo, (size_t)L->span * sizeof(double), stream);
    } else {
        fa_hip_memcpy_h2d(o + L->lo, re + L->lo, (size_t)L->span * sizeof(double), stream);
    }
}
static void stage_d2h(const stage_layout *L, double *st, double *re, i64 im, void *stream) {
    double *o = stage_origin(L, st);
    if (L->two) {
        fa_hip_memcpy_d2h(re + L->lo, o + L->lo, (size_t)L->span * sizeof(double), stream);
        fa_hip_memcpy_d2h(re + im + L->lo, o + L->im_dev + L->lo, (size_t)L->span * sizeof(double), stream);
    } else {
        fa_hip_memcpy_d2h(re + L->lo, o + L->lo, (size_t)L->span * sizeof(double), stream);
    }
}

static double *stage_buf(double **slot, size_t *have, size_t need) {
    if (*have < need) {
        fa_hip_free(*slot);
        *slot = (double *)fa_hip_malloc(need);
        if (!*slot) {
            fprintf(stderr, "fftw3_amd: no device memory to stage the host arrays of this execution\n");
            abort();
        }
        *have = need;
    }
    return *slot;
}

/* per-step HIP-event timing of one execution: p->prof_ms / p->prof_launches, set by
   fftw_amd_execute_profiled under the plan's lock (never process-wide: another plan's
   execute on another thread must not see them) */

static void fa_run_locked(plan *p, double *ri, double *ii, double *ro, double *io);

/* serialised per plan: concurrent fftw_execute calls on one plan are legal in the
   reference (A.c:433, re-entrant executors) but here they share scratch and streams */
void fa_run(plan *p, double *ri, double *ii, double *ro, double *io) {
    pthread_mutex_lock(&p->lock);
    /* New-array execution on arrays aligned differently from the plan's own (a caller error by FFTW's rules,
       fftw3.h "new-array execute"): the plan may hold steps whose kernels rely on the 16-byte alignment it
       saw at planning time and have no other executor.  Run the FFTW_UNALIGNED twin instead -- same problem,
       same tables' values, only kernels that serve every alignment -- rather than abort() inside a library. */
    if (!(p->flags & FFTW_UNALIGNED) && p->ri && p->ro && p->batch > 0 &&
        ((((size_t)ri ^ (size_t)p->ri) | ((size_t)ii ^ (size_t)p->ii) | ((size_t)ro ^ (size_t)p->ro) | ((size_t)io ^ (size_t)p->io)) & 15)) {
        if (!p->alt) p->alt = fa_unaligned_twin(p);
        if (p->alt) {
            p->alt->stream = p->stream;
            fa_run_locked(p->alt, ri, ii, ro, io);
            pthread_mutex_unlock(&p->lock);
            return;
        }
    }
    fa_run_locked(p, ri, ii, ro, io);
    pthread_mutex_unlock(&p->lock);
}

static void fa_run_locked(plan *p, double *ri, double *ii, double *ro, double *io) {
    double *bufs[FA_MAXBUF];
    void *tabs[FA_MAXTAB];
    int i, in_host = 0, out_host = 0, host_inplace = 0, hpipe = 0;
    i64 cs, nhchunks = 0, hp_ibs = 0, hp_obs = 0, hp_ispan = 0, hp_ospan = 0;
    void **ev_in = NULL, **ev_out = NULL;
    i64 in_im = ii - ri, out_im = io - ro, out_im_host = 0;
    stage_layout Lin, Lout;
    double *din = ri, *dout = ro;
    i64 in_span_lo = p->in_lo, in_span_hi = p->in_hi;
    i64 out_span_lo = p->out_lo, out_span_hi = p->out_hi;

    if (p->batch == 0) return;
    if (fa_device_init(p)) abort();

    if (FA_REAL_IN(p->type)) in_im = 0;
    if (FA_REAL_OUT(p->type)) out_im = 0;

    /* plain host arrays are staged through device copies (slow path, PCIe).  A split array whose
       imaginary plane lies beyond the real one (|im| >= span: typically two separate allocations,
       the usual fftw_plan_guru_split_dft call) is staged as two packed planes -- never as one span
       from the lower to the upper plane, which would read and write whatever the caller keeps
       between the two allocations. */
    in_host = !fa_hip_is_device_ptr(ri);
    out_host = !fa_hip_is_device_ptr(ro);
    Lin = stage_layout_of(in_span_lo, in_span_hi, in_im);
    Lout = stage_layout_of(out_span_lo, out_span_hi, out_im);
    host_inplace = in_host && out_host && ro == ri && (io == ii || FA_REAL_OUT(p->type) || FA_REAL_IN(p->type));
    if (host_inplace) {
        /* both views share one staging image: same plane layout, union of the spans */
        if (Lin.two != Lout.two || (Lin.two && (Lin.lo != Lout.lo || Lin.hi != Lout.hi || Lin.im_dev != Lout.im_dev))) {
            fprintf(stderr, "fftw3_amd: in-place execution on host arrays whose input and output views have "
                            "different split layouts is not supported\n");
            abort();
        }
        if (!Lin.two) {
            if (Lout.lo < Lin.lo) Lin.lo = Lout.lo;
            if (Lout.hi > Lin.hi) Lin.hi = Lout.hi;
            Lin.span = Lin.hi - Lin.lo + 1;
            Lout.lo = Lin.lo; Lout.hi = Lin.hi; Lout.span = Lin.span;
        }
    }
    /* Host pipeline: a batch that is dense on both sides (every transform's span below the batch stride) and
       runs in several chunks is staged chunk by chunk on two copy streams -- upload of chunk c+1, compute of
       chunk c and download of chunk c-1 overlap, so an unmodified host caller sees max(H2D, D2H) of the PCIe
       link instead of their sum plus the compute.  Everything else is staged whole, as before. */
    if (in_host && out_host && !host_inplace && !Lin.two && !Lout.two && p->hrank == 1 && !p->prof_ms &&
        p->chunk > 0 && p->chunk < p->batch && !p->single_chunk && !(p->nslots > 1 && p->pstream[0] && p->lanes <= 1)) {
        const i64 B = p->batch, ibs = p->hdims[0].is, obs = p->hdims[0].os;
        const i64 it_span = Lin.span - (B - 1) * ibs, ot_span = Lout.span - (B - 1) * obs;   /* one transform's span */
        static int hp_mode = -1;        /* FFTW_AMD_HOST_PIPELINE=0 switches it off */
        if (hp_mode < 0) { const char *e = getenv("FFTW_AMD_HOST_PIPELINE"); hp_mode = e ? atoi(e) : 1; }
        if (hp_mode && ibs > 0 && obs > 0 && it_span > 0 && ot_span > 0 && it_span <= ibs && ot_span <= obs &&
            Lout.span == p->out_written) {
            hpipe = 1;
            hp_ibs = ibs; hp_obs = obs; hp_ispan = it_span; hp_ospan = ot_span;
            nhchunks = (B + p->chunk - 1) / p->chunk;
            if (!p->hstream[0]) { p->hstream[0] = fa_hip_stream_create(); p->hstream[1] = fa_hip_stream_create(); }
            ev_in = (void **)calloc((size_t)nhchunks, sizeof(void *));
            ev_out = (void **)calloc((size_t)nhchunks, sizeof(void *));
        }
    }
    if (in_host) {
        double *st = stage_buf(&p->stage_in, &p->stage_in_bytes, stage_bytes(&Lin));
        din = stage_origin(&Lin, st);
        if (hpipe) {
            /* uploads are queued now, in chunk order, behind whatever the caller's stream already holds */
            i64 c;
            void *e0 = fa_hip_event_create();
            fa_hip_event_record(e0, p->stream);
            fa_hip_stream_wait_event(p->hstream[0], e0);
            fa_hip_stream_wait_event(p->hstream[1], e0);
            fa_hip_event_destroy(e0);
            for (c = 0; c < nhchunks; ++c) {
                const i64 c0 = c * p->chunk, cn = (p->batch - c0 < p->chunk) ? p->batch - c0 : p->chunk;
                const i64 lo = Lin.lo + c0 * hp_ibs, len = (cn - 1) * hp_ibs + hp_ispan;
                fa_hip_memcpy_h2d(din + lo, ri + lo, (size_t)len * sizeof(double), p->hstream[0]);
                ev_in[c] = fa_hip_event_create();
                fa_hip_event_record(ev_in[c], p->hstream[0]);
            }
        } else {
            stage_h2d(&Lin, st, ri, in_im, p->stream);
        }
        in_im = Lin.im_dev;
    }
    if (out_host) {
        if (host_inplace) {
            dout = din;
        } else {
            double *st = stage_buf(&p->stage_out, &p->stage_out_bytes, stage_bytes(&Lout));
            /* gaps between output elements must survive the round trip */
            if ((i64)(stage_bytes(&Lout) / sizeof(double)) != p->out_written)
                stage_h2d(&Lout, st, ro, out_im, p->stream);
            dout = stage_origin(&Lout, st);
        }
        out_im_host = out_im;
        out_im = Lout.im_dev;
    }

    bufs[0] = din;
    bufs[1] = dout;
    for (i = 2; i < p->nbufs; ++i) bufs[i] = p->dbuf[i];
    for (i = 0; i < p->ntabs; ++i) tabs[i] = p->tabs[i].dev;

    if (p->pair) {
        /* launch k: pass 2 of chunk k-1 (scratch slot (k-1) & 1) + pass 1 of chunk k (slot k & 1) */
        const i64 nchunks = (p->batch + p->chunk - 1) / p->chunk;
        i64 k;
        int ok = 1;
        for (k = 0; k <= nchunks && ok; ++k) {
            double *sb1[FA_MAXBUF], *sb2[FA_MAXBUF];
            fftw_amd_step_desc d1 = p->steps[0], d2 = p->steps[1];
            const i64 cs1 = k * p->chunk, cs2 = (k - 1) * p->chunk;
            const i64 cn1 = k < nchunks ? (p->batch - cs1 < p->chunk ? p->batch - cs1 : p->chunk) : 0;
            const i64 cn2 = k > 0 ? (p->batch - cs2 < p->chunk ? p->batch - cs2 : p->chunk) : 0;
            void *e0 = NULL, *e1 = NULL;
            sb1[0] = sb2[0] = bufs[0];
            sb1[1] = sb2[1] = bufs[1];
            sb1[2] = bufs[2] + (i64)(k & 1) * p->buf_reals[2];
            sb2[2] = bufs[2] + (i64)((k + 1) & 1) * p->buf_reals[2];
            if (d1.src_im == p->in_im && !FA_REAL_IN(p->type)) d1.src_im = in_im;
            if (d2.dst_im == p->out_im && !FA_REAL_OUT(p->type)) d2.dst_im = out_im;
            if (p->prof_ms) { e0 = fa_hip_event_create(); e1 = fa_hip_event_create(); fa_hip_event_record(e0, p->stream); }
            if (hpipe && k < nchunks) fa_hip_stream_wait_event(p->stream, ev_in[k]);     /* chunk k is on the device */
            if (fa_hip_launch_pair1024(&d2, sb2, cs2, cn2, &d1, sb1, cs1, cn1, tabs, p->stream)) {
                if (k == 0) { ok = 0; }          /* not the pair the kernel is built for: ordinary launches below */
                else abort();
            }
            if (hpipe && ok && k > 0) {
                /* chunk k-1 is complete: download it while the next launches run */
                const i64 lo = Lout.lo + cs2 * hp_obs, len = (cn2 - 1) * hp_obs + hp_ospan;
                ev_out[k - 1] = fa_hip_event_create();
                fa_hip_event_record(ev_out[k - 1], p->stream);
                fa_hip_stream_wait_event(p->hstream[1], ev_out[k - 1]);
                fa_hip_memcpy_d2h(ro + lo, dout + lo, (size_t)len * sizeof(double), p->hstream[1]);
            }
            if (p->prof_ms) {
                fa_hip_event_record(e1, p->stream);
                fa_hip_stream_sync(p->stream);
                if (ok) { p->prof_ms[0] += (double)fa_hip_event_elapsed_ms(e0, e1); p->prof_launches[0] += 1; }
                fa_hip_event_destroy(e0);
                fa_hip_event_destroy(e1);
            }
        }
        if (!ok) p->pair = 0;
    }
    if (!p->pair) {
        i64 nchunks = (p->batch + p->chunk - 1) / p->chunk, ev = 0, c = 0;
        void **events = NULL;
        const int pipe = p->nslots > 1 && p->pstream[0] && p->lanes <= 1;
        /* profiled executions keep the lanes: the HIP events around a launch sit on the lane's own stream and time
           the launch as it really runs, beside the other lane's (rocprofv3's kernel durations see the same) */
        const int lanes = p->lanes > 1 ? p->lanes : 1;
        if (p->prof_ms) events = (void **)malloc(sizeof(void *) * (size_t)(2 * nchunks * p->nsteps));
        if (pipe || lanes > 1) {
            /* the side streams start after everything already queued on the caller's stream */
            fa_hip_event_record(p->ev_begin, p->stream);
            for (i = 0; i < (pipe ? 2 : lanes); ++i) fa_hip_stream_wait_event(p->pstream[i], p->ev_begin);
        }
        for (cs = 0; cs < p->batch; cs += p->chunk, ++c) {
            i64 cn = p->batch - cs < p->chunk ? p->batch - cs : p->chunk;
            double *sb[FA_MAXBUF];
            int slot = pipe ? (int)(c % p->nslots) : (lanes > 1 ? (int)(c % lanes) : 0);
            void *lane_st = lanes > 1 ? p->pstream[c % lanes] : p->stream;
            sb[0] = bufs[0];
            sb[1] = bufs[1];
            for (i = 2; i < p->nbufs; ++i) sb[i] = bufs[i] + (i64)slot * p->buf_reals[i];
            if (hpipe) fa_hip_stream_wait_event(lane_st, ev_in[c]);
            for (i = 0; i < p->nsteps; ++i) {
                fftw_amd_step_desc d = p->steps[i];
                void *st = lane_st;
                if (pipe) {
                    st = p->pstream[i < p->split ? 0 : 1];
                    /* stage A reuses a slot only after stage B of its previous user is done */
                    if (i == 0 && c >= p->nslots) fa_hip_stream_wait_event(st, p->ev_b[slot]);
                    if (i == p->split) fa_hip_stream_wait_event(st, p->ev_a[slot]);
                }
                /* split-array callers may pass different re/im distances per call */
                if (d.src_buf == 0 && !FA_REAL_IN(p->type) && d.src_im == p->in_im) d.src_im = in_im;
                if (d.dst_buf == 1 && !FA_REAL_OUT(p->type) && d.dst_im == p->out_im) d.dst_im = out_im;
                if (events) { events[ev] = fa_hip_event_create(); fa_hip_event_record(events[ev++], st); }
                if (fa_hip_launch_step(&d, sb, tabs, cs, cn, st)) abort();
                if (events) { events[ev] = fa_hip_event_create(); fa_hip_event_record(events[ev++], st); }
                if (pipe && i == p->split - 1) fa_hip_event_record(p->ev_a[slot], st);
                if (pipe && i == p->nsteps - 1) fa_hip_event_record(p->ev_b[slot], st);
            }
            if (hpipe && !ev_out[c]) {
                const i64 lo = Lout.lo + cs * hp_obs, len = (cn - 1) * hp_obs + hp_ospan;
                ev_out[c] = fa_hip_event_create();
                fa_hip_event_record(ev_out[c], lane_st);
                fa_hip_stream_wait_event(p->hstream[1], ev_out[c]);
                fa_hip_memcpy_d2h(ro + lo, dout + lo, (size_t)len * sizeof(double), p->hstream[1]);
            }
        }
        if (pipe || lanes > 1) {
            /* the caller's stream continues after the side streams have drained */
            for (i = 0; i < (pipe ? 2 : lanes); ++i) {
                fa_hip_event_record(p->ev_end[i], p->pstream[i]);
                fa_hip_stream_wait_event(p->stream, p->ev_end[i]);
            }
        }
        if (events) {
            fa_hip_stream_sync(p->stream);
            ev = 0;
            for (c = 0; c < nchunks; ++c)
                for (i = 0; i < p->nsteps; ++i) {
                    p->prof_ms[i] += (double)fa_hip_event_elapsed_ms(events[ev], events[ev + 1]);
                    p->prof_launches[i] += 1;
                    fa_hip_event_destroy(events[ev]);
                    fa_hip_event_destroy(events[ev + 1]);
                    ev += 2;
                }
            free(events);
        }
    }

    if (hpipe) {
        i64 c;
        fa_hip_stream_sync(p->stream);
        fa_hip_stream_sync(p->hstream[1]);
        for (c = 0; c < nhchunks; ++c) {
            if (ev_in[c]) fa_hip_event_destroy(ev_in[c]);
            if (ev_out[c]) fa_hip_event_destroy(ev_out[c]);
        }
        free(ev_in);
        free(ev_out);
        return;
    }
    if (out_host) stage_d2h(&Lout, host_inplace ? p->stage_in : p->stage_out, ro, out_im_host, p->stream);
    if (in_host || out_host) fa_hip_stream_sync(p->stream);
}

/* One execution on the plan's own arrays with a HIP event pair around every
   launch (on the stream the kernels run on).  ms[i] / launches[i] accumulate
   the time and the number of launches of step i.  Returns the step count. */
int fftw_amd_execute_profiled(fftw_plan p, double *ms, long long *launches, int cap) {
    int i;
    if (!p || cap < p->nsteps) return -1;
    for (i = 0; i < p->nsteps; ++i) { ms[i] = 0.0; launches[i] = 0; }
    pthread_mutex_lock(&p->lock);
    p->prof_ms = ms;
    p->prof_launches = launches;
    fa_run_locked(p, p->ri, p->ii, p->ro, p->io);
    p->prof_ms = NULL;
    p->prof_launches = NULL;
    pthread_mutex_unlock(&p->lock);
    return p->nsteps;
}

/* ------------------------------------------------------------- printing */

static const char *kind_name(int k) {
    switch (k) {
    case FFTW_AMD_STEP_PASS: return "pass";
    case FFTW_AMD_STEP_COPY: return "copy";
    case FFTW_AMD_STEP_R2C_POST: return "r2c-untangle";
    case FFTW_AMD_STEP_C2R_PRE: return "c2r-tangle";
    case FFTW_AMD_STEP_RADER_MUL: return "rader-mul";
    case FFTW_AMD_STEP_HERM_EXPAND: return "herm-expand";
    case FFTW_AMD_STEP_R2C_POST4: return "r2c-untangle4";
    case FFTW_AMD_STEP_C2R_PRE4: return "c2r-tangle4";
    case FFTW_AMD_STEP_R2R: return "r2r";
    }
    return "?";
}

/* bounded append: never writes past cap, whatever the numbers print to */
static void sapp(char *s, size_t cap, size_t *len, const char *fmt, ...) {
    va_list ap;
    int w;
    if (*len + 1 >= cap) return;
    va_start(ap, fmt);
    w = vsnprintf(s + *len, cap - *len, fmt, ap);
    va_end(ap);
    if (w < 0) return;
    *len += (size_t)w;
    if (*len >= cap) *len = cap - 1;
}

/* lisp-like dump in the spirit of the reference's fftw_print_plan
   (fftw/fftw_api.c:1046-1080) */
char *fa_sprint(const plan *p) {
    size_t cap = 256 + (size_t)p->nsteps * 512, len = 0;
    char *s = (char *)malloc(cap);
    if (!s) return NULL;
    s[0] = 0;
    int i, j;
    const char *tn = p->type == FA_C2C ? "dft" : p->type == FA_R2C ? "rdft2-r2c" : p->type == FA_C2R ? "rdft2-c2r" : "rdft-r2r";
    sapp(s, cap, &len, "(hip-%s batch=%lld chunk=%lld", tn, p->batch, p->chunk);
    for (i = 0; i < p->nsteps; ++i) {
        const fftw_amd_step_desc *d = &p->steps[i];
        sapp(s, cap, &len, "\n  (%s", kind_name(d->kind));
        if (d->kind == FFTW_AMD_STEP_PASS) {
            /* which kernel runs the pass: reg32x32 / reg2 / reg3 = register-resident
               (pass1024 / passrr / pass3s), lds = runtime-radix LDS kernel + its radices */
            sapp(s, cap, &len, "-%d/", d->L);
            if (d->variant == FFTW_AMD_K_R2C) sapp(s, cap, &len, d->aux_valid ? "r2c-rows+r2r-post" : "r2c-rows");
            else if (d->variant == FFTW_AMD_K_C2R) sapp(s, cap, &len, d->aux_valid ? (d->aux_buf > 0 ? "c2r-rows+r2r-pre+post" : "c2r-rows+r2r-pre") : "c2r-rows");
            else if (d->variant == FFTW_AMD_K_P1024) sapp(s, cap, &len, "reg32x32");
            else if (d->variant == FFTW_AMD_K_RR) sapp(s, cap, &len, "reg2");
            else if (d->variant == FFTW_AMD_K_R3 && (d->flags & FFTW_AMD_F_REAL_DEC)) sapp(s, cap, &len, "reg3+real-decimated");
            else if (d->variant == FFTW_AMD_K_R3) sapp(s, cap, &len, (d->flags & FFTW_AMD_F_LO_DFT) ? (d->tile_lo_n == 4 ? "reg3+dft4-across-rows" : "reg3+dft2-across-rows") : "reg3");
            else if (d->variant == FFTW_AMD_K_R1) sapp(s, cap, &len, "reg1");
            else if (d->variant == FFTW_AMD_K_BLUE) sapp(s, cap, &len, "bluestein-rows n=%lld", (long long)d->aux_n);
            else {
                sapp(s, cap, &len, "lds:");
                for (j = 0; j < d->nradices; ++j)
                    sapp(s, cap, &len, "%s%d", j ? "x" : "", d->radices[j]);
            }
            sapp(s, cap, &len, " tile=%d", d->tile);
            if (d->tw_n) sapp(s, cap, &len, " tw=%lld", d->tw_n);
        } else if (d->kind == FFTW_AMD_STEP_R2R ||
                   ((d->kind == FFTW_AMD_STEP_R2C_POST || d->kind == FFTW_AMD_STEP_R2C_POST4 ||
                     d->kind == FFTW_AMD_STEP_C2R_PRE || d->kind == FFTW_AMD_STEP_C2R_PRE4) && d->variant)) {
            /* untangle / tangle steps carry the name of a fused r2r epilogue / prologue */
            static const char *mn[] = { "?", "pre-hc2r", "pre-e10", "pre-o10", "pre-e01", "pre-o01", "pre-e00",
                "pre-o00", "pre-e11", "pre-o11", "pre-e11odd", "pre-o11odd", "post-r2hc", "post-dht",
                "post-e10", "post-o10", "post-e01", "post-o01", "post-e00", "post-o00", "post-e11",
                "post-o11", "post-e11odd", "post-o11odd" };
            int m = (d->variant >= 1 && d->variant <= FFTW_AMD_R2R_POST_O11ODD) ? d->variant : 0;
            sapp(s, cap, &len, "%s%s n=%lld", d->kind == FFTW_AMD_STEP_R2R ? "-" : "+r2r-",
                                    mn[m], d->aux_n);
        } else {
            sapp(s, cap, &len, " n=%lld", d->aux_n);
        }
        sapp(s, cap, &len, " buf%d->buf%d x", d->src_buf, d->dst_buf);
        for (j = 0; j < d->ndims; ++j)
            sapp(s, cap, &len, "%s%lld", j ? "," : "", d->dim_n[j]);
        sapp(s, cap, &len, ")");
    }
    sapp(s, cap, &len, ")");
    return s;
}
