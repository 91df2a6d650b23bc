/*
 * sharded.c -- one batched transform sharded over several GPUs of a node, behind the C ABI.
 *
 * Whole batch elements are independent, so the batch is cut by the block rule of the
 * reference's own parallel layers -- worker g of P owns [g*ceil(B/P), min(B, (g+1)*ceil(B/P)))
 * (fftw/threads/dft-vrank-geq1.c:158-159 block_size = (vl + nthr - 1) / nthr, and
 * fftw/mpi/block.c:35-42) -- and every device gets what the reference gives every thread
 * (dft-vrank-geq1.c:140-175: the vector loop split across workers inside the library):
 * one plan replica with its own twiddle tables and scratch, one stream, one host thread.
 * Nothing is exchanged inside a transform.  The optional all-gather of the output shards
 * (north_star: "RCCL all-gather over xGMI to reassemble output") runs after the transforms,
 * on the same streams: through RCCL (librccl.so, loaded on first use) when every shard sits on
 * a different device, as direct peer-to-peer pushes otherwise.
 */
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fa_plan.h"
#include "fa_hip.h"

#define FA_MAXDEV 32

enum { CMD_NONE = 0, CMD_EXEC, CMD_SYNC, CMD_GATHER, CMD_EXIT };

typedef struct {
    struct fftw_amd_sharded_plan_s *owner;
    int g;
    pthread_t thr;
    int started;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int cmd;            /* posted command, CMD_NONE when idle */
    int busy;
} shard_worker;

struct fftw_amd_sharded_plan_s {
    int type;                      /* FA_C2C / FA_R2C / FA_C2R */
    int ndev;
    int devs[FA_MAXDEV];
    long long howmany;
    long long lo[FA_MAXDEV], hi[FA_MAXDEV];
    fftw_plan replica[FA_MAXDEV];  /* NULL for an empty shard */
    void *stream[FA_MAXDEV];
    void *in[FA_MAXDEV], *out[FA_MAXDEV];
    long long out_dist_bytes;      /* bytes between consecutive transforms of the output */
    size_t out_elem_bytes;         /* bytes one transform's output spans (dense layouts only) */
    int dense_out;
    shard_worker w[FA_MAXDEV];
    void *const *gather_dst;       /* argument of the pending CMD_GATHER */
    /* RCCL, when used (the library itself is process-wide: g_nccl) */
    void *comms[FA_MAXDEV];
    int rccl_ready;
};

void fftw_amd_shard_range(long long howmany, int nshards, int g, long long *lo, long long *hi) {
    long long blk = nshards > 0 ? (howmany + nshards - 1) / nshards : howmany;
    long long a = blk * g, b = blk * (g + 1);
    if (a > howmany) a = howmany;
    if (b > howmany) b = howmany;
    if (lo) *lo = a;
    if (hi) *hi = b;
}

static void run_cmd(struct fftw_amd_sharded_plan_s *p, int g, int cmd) {
    switch (cmd) {
    case CMD_EXEC:
        if (p->replica[g]) {
            if (p->type == FA_R2C) fftw_execute_dft_r2c(p->replica[g], (double *)p->in[g], (fftw_complex *)p->out[g]);
            else if (p->type == FA_C2R) fftw_execute_dft_c2r(p->replica[g], (fftw_complex *)p->in[g], (double *)p->out[g]);
            else fftw_execute_dft(p->replica[g], (fftw_complex *)p->in[g], (fftw_complex *)p->out[g]);
        }
        break;
    case CMD_SYNC:
        fa_hip_stream_sync(p->stream[g]);
        break;
    case CMD_GATHER: {
        /* source-driven full mesh: this shard is pushed into every destination image, ordered
           behind the transforms on this shard's stream */
        int d;
        size_t bytes = (size_t)(p->hi[g] - p->lo[g]) * (size_t)p->out_dist_bytes;
        if (!bytes) break;
        for (d = 0; d < p->ndev; ++d) {
            char *dst = (char *)p->gather_dst[d] + (size_t)p->lo[g] * (size_t)p->out_dist_bytes;
            if (dst == (char *)p->out[g]) continue;          /* the shard already lives inside the image */
            fa_hip_memcpy_peer(dst, p->devs[d], p->out[g], p->devs[g], bytes, p->stream[g]);
        }
        break;
    }
    default: break;
    }
}

static void *worker_main(void *arg) {
    shard_worker *w = (shard_worker *)arg;
    struct fftw_amd_sharded_plan_s *p = w->owner;
    const int g = w->g;
    fa_hip_set_device(p->devs[g]);                 /* everything this thread allocates or launches is on its device */
    p->stream[g] = fa_hip_stream_create();
    if (p->replica[g]) fftw_amd_plan_set_stream(p->replica[g], p->stream[g]);
    pthread_mutex_lock(&w->mu);
    w->busy = 0;
    pthread_cond_broadcast(&w->cv);
    for (;;) {
        int cmd;
        while (w->cmd == CMD_NONE) pthread_cond_wait(&w->cv, &w->mu);
        cmd = w->cmd;
        pthread_mutex_unlock(&w->mu);
        if (cmd != CMD_EXIT) run_cmd(p, g, cmd);
        pthread_mutex_lock(&w->mu);
        w->cmd = CMD_NONE;
        w->busy = 0;
        pthread_cond_broadcast(&w->cv);
        if (cmd == CMD_EXIT) break;
    }
    pthread_mutex_unlock(&w->mu);
    return NULL;
}

/* post a command to every worker, then wait until every worker has carried it out */
static void broadcast_cmd(struct fftw_amd_sharded_plan_s *p, int cmd) {
    int g;
    for (g = 0; g < p->ndev; ++g) {
        shard_worker *w = &p->w[g];
        pthread_mutex_lock(&w->mu);
        while (w->busy) pthread_cond_wait(&w->cv, &w->mu);
        w->cmd = cmd;
        w->busy = 1;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
    }
    for (g = 0; g < p->ndev; ++g) {
        shard_worker *w = &p->w[g];
        pthread_mutex_lock(&w->mu);
        while (w->busy) pthread_cond_wait(&w->cv, &w->mu);
        pthread_mutex_unlock(&w->mu);
    }
}

static int start_workers(struct fftw_amd_sharded_plan_s *p) {
    int g;
    if (fa_hip_device_count() <= 0) {
        fprintf(stderr, "fftw3_amd: no HIP device available: a sharded plan cannot execute (no CPU fallback)\n");
        return -1;
    }
    for (g = 0; g < p->ndev; ++g) {
        shard_worker *w = &p->w[g];
        if (w->started) continue;
        if (p->devs[g] < 0 || p->devs[g] >= fa_hip_device_count()) {
            fprintf(stderr, "fftw3_amd: sharded plan names device %d, but only %d are visible\n", p->devs[g], fa_hip_device_count());
            return -1;
        }
        w->owner = p; w->g = g; w->cmd = CMD_NONE; w->busy = 1;
        pthread_mutex_init(&w->mu, NULL);
        pthread_cond_init(&w->cv, NULL);
        if (pthread_create(&w->thr, NULL, worker_main, w)) return -1;
        w->started = 1;
    }
    for (g = 0; g < p->ndev; ++g) {
        shard_worker *w = &p->w[g];
        pthread_mutex_lock(&w->mu);
        while (w->busy) pthread_cond_wait(&w->cv, &w->mu);
        pthread_mutex_unlock(&w->mu);
    }
    return 0;
}

static struct fftw_amd_sharded_plan_s *mk_sharded(int type, int rank, const int *n, int howmany, int ndev, const int *devs,
                                                 void *const *in, const int *inembed, int istride, int idist,
                                                 void *const *out, const int *onembed, int ostride, int odist,
                                                 int sign, unsigned flags) {
    struct fftw_amd_sharded_plan_s *p;
    int g, i, ndevices, saved_dev;
    long long out_elems = 1;
    if (ndev < 1 || ndev > FA_MAXDEV || !in || !out || howmany < 0 || rank < 1 || !n) return NULL;
    p = (struct fftw_amd_sharded_plan_s *)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->type = type; p->ndev = ndev; p->howmany = howmany;
    for (i = 0; i < rank; ++i) {
        long long e = (onembed ? onembed[i] : n[i]);
        if (!onembed && type == FA_R2C && i == rank - 1) e = n[i] / 2 + 1;
        out_elems *= e;
    }
    p->out_dist_bytes = (long long)odist * (type == FA_C2R ? 8 : 16);
    p->out_elem_bytes = (size_t)out_elems * (size_t)ostride * (type == FA_C2R ? 8 : 16);
    p->dense_out = odist > 0 && (long long)p->out_elem_bytes <= p->out_dist_bytes;
    ndevices = fa_hip_device_count();
    saved_dev = ndevices > 0 ? fa_hip_get_device() : -1;
    for (g = 0; g < ndev; ++g) {
        p->devs[g] = devs ? devs[g] : g;
        /* with a device runtime present every named device must exist (without one the plan can
           still be built and inspected: the CPU test tier does) */
        if (ndevices > 0 && (p->devs[g] < 0 || p->devs[g] >= ndevices)) {
            fprintf(stderr, "fftw3_amd: sharded plan names device %d, but only %d are visible\n", p->devs[g], ndevices);
            free(p);
            return NULL;
        }
    }
    for (g = 0; g < ndev; ++g) {
        long long cnt;
        fftw_amd_shard_range(howmany, ndev, g, &p->lo[g], &p->hi[g]);
        cnt = p->hi[g] - p->lo[g];
        p->in[g] = in[g]; p->out[g] = out[g];
        if (cnt <= 0) continue;
        if (!in[g] || !out[g]) { if (saved_dev >= 0) fa_hip_set_device(saved_dev); fftw_amd_destroy_sharded_plan(p); return NULL; }
        /* the replica is an ordinary plan of this library over the shard's own batch.  A plan puts its
           tables and scratch on the CURRENT device when it is created (api.c finish_locked ->
           fa_device_init) and FFTW_MEASURE times candidates there: make that the shard's device */
        if (saved_dev >= 0) fa_hip_set_device(p->devs[g]);
        if (type == FA_R2C)
            p->replica[g] = fftw_plan_many_dft_r2c(rank, n, (int)cnt, (double *)in[g], inembed, istride, idist,
                                                   (fftw_complex *)out[g], onembed, ostride, odist, flags);
        else if (type == FA_C2R)
            p->replica[g] = fftw_plan_many_dft_c2r(rank, n, (int)cnt, (fftw_complex *)in[g], inembed, istride, idist,
                                                   (double *)out[g], onembed, ostride, odist, flags);
        else
            p->replica[g] = fftw_plan_many_dft(rank, n, (int)cnt, (fftw_complex *)in[g], inembed, istride, idist,
                                               (fftw_complex *)out[g], onembed, ostride, odist, sign, flags);
        if (!p->replica[g]) { if (saved_dev >= 0) fa_hip_set_device(saved_dev); fftw_amd_destroy_sharded_plan(p); return NULL; }
    }
    if (saved_dev >= 0) fa_hip_set_device(saved_dev);
    return p;
}

fftw_amd_sharded_plan fftw_amd_plan_many_dft_sharded(int rank, const int *n, int howmany, int ndev, const int *devs,
                                                     fftw_complex *const *in, const int *inembed, int istride, int idist,
                                                     fftw_complex *const *out, const int *onembed, int ostride, int odist,
                                                     int sign, unsigned flags) {
    return mk_sharded(FA_C2C, rank, n, howmany, ndev, devs, (void *const *)in, inembed, istride, idist,
                      (void *const *)out, onembed, ostride, odist, sign, flags);
}
fftw_amd_sharded_plan fftw_amd_plan_many_dft_r2c_sharded(int rank, const int *n, int howmany, int ndev, const int *devs,
                                                         double *const *in, const int *inembed, int istride, int idist,
                                                         fftw_complex *const *out, const int *onembed, int ostride, int odist,
                                                         unsigned flags) {
    return mk_sharded(FA_R2C, rank, n, howmany, ndev, devs, (void *const *)in, inembed, istride, idist,
                      (void *const *)out, onembed, ostride, odist, FFTW_FORWARD, flags);
}
fftw_amd_sharded_plan fftw_amd_plan_many_dft_c2r_sharded(int rank, const int *n, int howmany, int ndev, const int *devs,
                                                         fftw_complex *const *in, const int *inembed, int istride, int idist,
                                                         double *const *out, const int *onembed, int ostride, int odist,
                                                         unsigned flags) {
    return mk_sharded(FA_C2R, rank, n, howmany, ndev, devs, (void *const *)in, inembed, istride, idist,
                      (void *const *)out, onembed, ostride, odist, FFTW_BACKWARD, flags);
}

int fftw_amd_sharded_num_shards(const fftw_amd_sharded_plan p) { return p ? p->ndev : 0; }
int fftw_amd_sharded_device(const fftw_amd_sharded_plan p, int g) { return (p && g >= 0 && g < p->ndev) ? p->devs[g] : -1; }
void fftw_amd_sharded_range(const fftw_amd_sharded_plan p, int g, long long *lo, long long *hi) {
    if (!p || g < 0 || g >= p->ndev) { if (lo) *lo = 0; if (hi) *hi = 0; return; }
    if (lo) *lo = p->lo[g];
    if (hi) *hi = p->hi[g];
}
fftw_plan fftw_amd_sharded_replica(const fftw_amd_sharded_plan p, int g) { return (p && g >= 0 && g < p->ndev) ? p->replica[g] : NULL; }

/* launches every shard's transforms (each on its own device, stream and host thread) and returns
   once everything is enqueued; fftw_amd_sharded_sync waits for the devices */
void fftw_amd_execute_sharded(const fftw_amd_sharded_plan p) {
    if (!p || p->howmany == 0) return;
    if (start_workers(p)) abort();
    broadcast_cmd(p, CMD_EXEC);
}
void fftw_amd_sharded_sync(const fftw_amd_sharded_plan p) {
    int g, any = 0;
    if (!p) return;
    for (g = 0; g < p->ndev; ++g) any |= p->w[g].started;
    if (any) broadcast_cmd(p, CMD_SYNC);
}

/* ---- RCCL, loaded once per process: only the entry points the gather needs ---- */
typedef int (*nccl_init_all_fn)(void **comms, int ndev, const int *devlist);
typedef int (*nccl_destroy_fn)(void *comm);
typedef int (*nccl_group_fn)(void);
typedef int (*nccl_bcast_fn)(const void *send, void *recv, size_t count, int dtype, int root, void *comm, void *stream);
typedef int (*nccl_allgather_fn)(const void *send, void *recv, size_t sendcount, int dtype, void *comm, void *stream);
typedef const char *(*nccl_errstr_fn)(int);
static struct {
    void *lib;
    nccl_init_all_fn init_all; nccl_destroy_fn destroy; nccl_group_fn gstart, gend; nccl_bcast_fn bcast;
    nccl_allgather_fn allgather; nccl_errstr_fn errstr;
    int nsyms;          /* entry points resolved */
    int ok;
} g_nccl;
static pthread_once_t g_nccl_once = PTHREAD_ONCE_INIT;
#define FA_NCCL_INT8 0     /* ncclInt8 / ncclChar: the gather moves opaque bytes */
#define FA_NCCL_NSYMS 7

/* The library is opened once and stays open for the life of the process (plans share the table; a
   dlclose while another plan's communicators are alive would pull the code from under them).
   FFTW_AMD_RCCL_LIB names another library with the same entry points (the CPU test tier uses a
   recording mock). */
static void rccl_load_once(void) {
    const char *over = getenv("FFTW_AMD_RCCL_LIB");
    void *h = NULL;
    if (over && *over) h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    else {
        h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) return;
    g_nccl.init_all = (nccl_init_all_fn)dlsym(h, "ncclCommInitAll");
    g_nccl.destroy = (nccl_destroy_fn)dlsym(h, "ncclCommDestroy");
    g_nccl.gstart = (nccl_group_fn)dlsym(h, "ncclGroupStart");
    g_nccl.gend = (nccl_group_fn)dlsym(h, "ncclGroupEnd");
    g_nccl.bcast = (nccl_bcast_fn)dlsym(h, "ncclBroadcast");
    g_nccl.allgather = (nccl_allgather_fn)dlsym(h, "ncclAllGather");
    g_nccl.errstr = (nccl_errstr_fn)dlsym(h, "ncclGetErrorString");
    g_nccl.nsyms = !!g_nccl.init_all + !!g_nccl.destroy + !!g_nccl.gstart + !!g_nccl.gend + !!g_nccl.bcast +
                   !!g_nccl.allgather + !!g_nccl.errstr;
    if (g_nccl.nsyms != FA_NCCL_NSYMS) { dlclose(h); return; }
    g_nccl.lib = h;
    g_nccl.ok = 1;
}

/* number of RCCL entry points the loader resolved (7 when the gather can use RCCL, 0 when the
   library cannot be opened); opens the library but makes no RCCL call */
int fftw_amd_rccl_probe(void) {
    pthread_once(&g_nccl_once, rccl_load_once);
    return g_nccl.ok ? g_nccl.nsyms : 0;
}

static int rccl_setup(struct fftw_amd_sharded_plan_s *p) {
    int g, h, rc;
    if (p->rccl_ready) return p->rccl_ready > 0 ? 0 : -1;
    p->rccl_ready = -1;
    for (g = 0; g < p->ndev; ++g)
        for (h = 0; h < g; ++h)
            if (p->devs[g] == p->devs[h]) return -1;       /* RCCL wants one rank per device */
    pthread_once(&g_nccl_once, rccl_load_once);
    if (!g_nccl.ok) return -1;
    rc = g_nccl.init_all(p->comms, p->ndev, p->devs);
    if (rc != 0) {
        fprintf(stderr, "fftw3_amd: ncclCommInitAll failed (%s); the gather uses peer-to-peer copies\n", g_nccl.errstr(rc));
        memset(p->comms, 0, sizeof(p->comms));
        return -1;
    }
    p->rccl_ready = 1;
    return 0;
}

/* The RCCL calls of one gather, in issue order (all inside ONE group).  Equal shards (the common
   case: ndev divides howmany): one ncclAllGather per rank, in place when the shard already sits at
   its slot of the image.  Ragged tails of the block rule (the last shards shorter or empty) cannot
   be expressed with equal counts: one ncclBroadcast per non-empty shard and rank instead.
   Each op is 6 values: kind (0 all-gather, 1 broadcast), rank d, root g (-1 for all-gather),
   send pointer, receive pointer, bytes.  Returns the number of ops (computed even if cap is short). */
int fftw_amd_sharded_gather_ops(const fftw_amd_sharded_plan p, void *const *full, long long *ops, int cap) {
    int g, d, n = 0, equal = 1;
    if (!p || !full) return -1;
    for (g = 1; g < p->ndev; ++g)
        if (p->hi[g] - p->lo[g] != p->hi[0] - p->lo[0]) equal = 0;
    if (p->hi[0] - p->lo[0] <= 0) equal = 0;
    if (equal) {
        const long long bytes = (p->hi[0] - p->lo[0]) * p->out_dist_bytes;
        for (d = 0; d < p->ndev; ++d, ++n) {
            if (n >= cap) continue;
            ops[6 * n + 0] = 0; ops[6 * n + 1] = d; ops[6 * n + 2] = -1;
            ops[6 * n + 3] = (long long)(size_t)p->out[d];
            ops[6 * n + 4] = (long long)(size_t)full[d];
            ops[6 * n + 5] = bytes;
        }
        return n;
    }
    for (g = 0; g < p->ndev; ++g) {
        const long long bytes = (p->hi[g] - p->lo[g]) * p->out_dist_bytes;
        if (bytes <= 0) continue;
        for (d = 0; d < p->ndev; ++d, ++n) {
            if (n >= cap) continue;
            ops[6 * n + 0] = 1; ops[6 * n + 1] = d; ops[6 * n + 2] = g;
            ops[6 * n + 3] = (long long)(size_t)p->out[g];       /* read on the root only */
            ops[6 * n + 4] = (long long)(size_t)((char *)full[d] + (size_t)p->lo[g] * (size_t)p->out_dist_bytes);
            ops[6 * n + 5] = bytes;
        }
    }
    return n;
}

/* Reassemble the output: full[d] is a device buffer on shard d's device with room for all `howmany`
   transforms (howmany * odist elements); on return (after fftw_amd_sharded_sync) each holds every
   shard's output at its place.  Requires the dense batch layout (odist >= one transform's span).
   mode: 0 = RCCL when possible, else peer-to-peer; 1 = peer-to-peer pushes; 2 = RCCL or fail.
   Returns 0 (peer-to-peer used), 1 (RCCL used), -1 on error. */
int fftw_amd_sharded_all_gather(const fftw_amd_sharded_plan p, void *const *full, int mode) {
    int g, d;
    if (!p || !full || !p->dense_out) return -1;
    if (p->howmany == 0) return 0;
    if (start_workers(p)) return -1;
    if (mode != 1 && rccl_setup(p) == 0) {
        /* Rank d's calls go on stream[d], the stream shard d's transforms were enqueued on
           (fftw_amd_execute_sharded returns after every worker has enqueued): each rank's part of
           the collective is ordered behind its own transforms by the stream, and the collective
           itself makes the receivers wait for the senders -- no host-side drain. */
        long long ops[6 * FA_MAXDEV * FA_MAXDEV];
        const int nops = fftw_amd_sharded_gather_ops(p, full, ops, FA_MAXDEV * FA_MAXDEV);
        int cur = fa_hip_get_device(), rc = 0, k;
        rc |= g_nccl.gstart();
        for (k = 0; k < nops; ++k) {
            const long long *o = ops + 6 * k;
            d = (int)o[1];
            fa_hip_set_device(p->devs[d]);
            if (o[0] == 0)
                rc |= g_nccl.allgather((const void *)(size_t)o[3], (void *)(size_t)o[4], (size_t)o[5], FA_NCCL_INT8,
                                       p->comms[d], p->stream[d]);
            else
                rc |= g_nccl.bcast((const void *)(size_t)o[3], (void *)(size_t)o[4], (size_t)o[5], FA_NCCL_INT8, (int)o[2],
                                   p->comms[d], p->stream[d]);
        }
        rc |= g_nccl.gend();
        fa_hip_set_device(cur);
        if (rc != 0) { fprintf(stderr, "fftw3_amd: RCCL gather failed (%s)\n", g_nccl.errstr(rc)); return -1; }
        return 1;
    }
    if (mode == 2) return -1;
    for (g = 0; g < p->ndev; ++g)
        for (d = 0; d < p->ndev; ++d) fa_hip_enable_peer(p->devs[g], p->devs[d]);
    p->gather_dst = full;
    broadcast_cmd(p, CMD_GATHER);
    return 0;
}

void fftw_amd_destroy_sharded_plan(fftw_amd_sharded_plan p) {
    int g;
    if (!p) return;
    for (g = 0; g < p->ndev; ++g) {
        shard_worker *w = &p->w[g];
        if (!w->started) continue;
        pthread_mutex_lock(&w->mu);
        while (w->busy) pthread_cond_wait(&w->cv, &w->mu);
        w->cmd = CMD_SYNC; w->busy = 1;
        pthread_cond_broadcast(&w->cv);
        while (w->busy) pthread_cond_wait(&w->cv, &w->mu);
        w->cmd = CMD_EXIT; w->busy = 1;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
        pthread_join(w->thr, NULL);
        pthread_mutex_destroy(&w->mu);
        pthread_cond_destroy(&w->cv);
    }
    if (p->rccl_ready > 0 && g_nccl.ok)
        for (g = 0; g < p->ndev; ++g) if (p->comms[g]) g_nccl.destroy(p->comms[g]);
    for (g = 0; g < p->ndev; ++g) {
        if (p->replica[g]) {
            /* tables and scratch were allocated on the shard's device */
            int cur = fa_hip_device_count() > 0 ? fa_hip_get_device() : 0;
            if (fa_hip_device_count() > 0) fa_hip_set_device(p->devs[g]);
            fftw_destroy_plan(p->replica[g]);
            if (fa_hip_device_count() > 0) fa_hip_set_device(cur);
        }
        if (p->stream[g]) {
            int cur = fa_hip_get_device();
            fa_hip_set_device(p->devs[g]);
            fa_hip_stream_destroy(p->stream[g]);
            fa_hip_set_device(cur);
        }
    }
    free(p);
}
