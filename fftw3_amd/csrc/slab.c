/*
 * slab.c -- ONE two- or three-dimensional complex transform spread over several GPUs of a node, behind the C ABI.
 *
 * Mirror of the reference's distributed-memory layer for the c2c case (fftw/mpi/fftw3-mpi.h:74-215:
 * fftw_mpi_local_size_2d / _3d, fftw_mpi_plan_dft_2d / _3d, fftw_mpi_execute_dft), with the MPI communicator
 * replaced by a list of devices of one process and the per-rank pointers by arrays of per-device pointers.  The
 * data distribution is the reference's: slabs along the first dimension by the block rule
 * (fftw/mpi/block.c:39-50: device g owns rows [g * ceil(n0 / P), ...)), normal order in, normal order out.
 *
 * Pipeline (the transposed-layout pipeline of fftw/mpi/dft-rank-geq2-transposed.c, with the global transposes
 * of fftw/mpi/transpose-alltoall.c done as peer-to-peer 2-D copies over xGMI -- no pack / unpack passes, and no
 * local transposes either: the second local transform simply runs down the strided first dimension):
 *
 *   1  every device g: transform over the trailing dimension(s) of its rows          in[g]  -> out[g]
 *   2  exchange: the column block of device r of every device's rows                 out[g] -> W[r]   ([n0][w_r])
 *   3  every device r: transforms of length n0 down its column block (stride w_r)    W[r] in place
 *   4  exchange back: rows of device g from every column block                       W[r]   -> out[g]
 *
 * with R = n1 (or n1 * n2) elements per row, column blocks cut on n1 by the same block rule (w_r = local_n1(r) *
 * n2).  Every local transform is an ordinary plan of this library created with its device current; every copy is
 * a hipMemcpy2DAsync on the receiving device's stream behind an event of the sending one.  One host thread
 * enqueues everything; fftw_amd_slab_sync waits.  fftw3_amd/slab.py is the multi-process form of the same layer
 * (torch.distributed in the place of MPI; r2c / c2r / r2r and the TRANSPOSED_IN / OUT layouts live there).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fa_plan.h"
#include "fa_hip.h"

#define FA_SLAB_MAXDEV 32

struct fftw_amd_slab_plan_s {
    int ndev, devs[FA_SLAB_MAXDEV];
    long long n0, n1, inner, R;                       /* rows, split dimension, elements per n1 entry, R = n1 * inner */
    long long lo0[FA_SLAB_MAXDEV], ln0[FA_SLAB_MAXDEV];   /* rows of device g */
    long long lo1[FA_SLAB_MAXDEV], ln1[FA_SLAB_MAXDEV];   /* its block of n1 */
    fftw_complex *in[FA_SLAB_MAXDEV], *out[FA_SLAB_MAXDEV];
    fftw_complex *W[FA_SLAB_MAXDEV];                  /* [n0][w_r], owned */
    fftw_plan rows[FA_SLAB_MAXDEV], cols[FA_SLAB_MAXDEV];
    void *stream[FA_SLAB_MAXDEV];
    void *ev_rows[FA_SLAB_MAXDEV], *ev_x1[FA_SLAB_MAXDEV], *ev_cols[FA_SLAB_MAXDEV], *ev_done[FA_SLAB_MAXDEV];
    int ran;                                           /* ev_done holds the end of a previous execution */
};

/* block rule of fftw/mpi/block.c:39-50 (default block = ceil(n / P)) */
static void slab_block(long long n, int P, int g, long long *lo, long long *len) {
    long long blk = (n + P - 1) / P, a = blk * g, b = blk * (g + 1);
    if (a > n) a = n;
    if (b > n) b = n;
    *lo = a;
    *len = b - a;
}

/* fftw_mpi_local_size_2d / _3d (fftw3-mpi.h:98-108): rows [*local_0_start, + *local_n0) of device g of ndev; the
   return value is the number of complex elements its in / out arrays must hold */
long long fftw_amd_slab_local_size(int rank, const long long *n, int ndev, int g, long long *local_n0, long long *local_0_start) {
    long long lo = 0, len = 0, rest = 1;
    int i;
    if (rank < 2 || rank > 3 || !n || ndev < 1 || g < 0 || g >= ndev) return -1;
    for (i = 1; i < rank; ++i) rest *= n[i];
    slab_block(n[0], ndev, g, &lo, &len);
    if (local_n0) *local_n0 = len;
    if (local_0_start) *local_0_start = lo;
    return len * rest;
}

void fftw_amd_destroy_slab_plan(struct fftw_amd_slab_plan_s *p) {
    int g, cur;
    if (!p) return;
    cur = fa_hip_device_count() > 0 ? fa_hip_get_device() : -1;
    for (g = 0; g < p->ndev; ++g) {
        if (cur >= 0) fa_hip_set_device(p->devs[g]);
        if (p->stream[g]) fa_hip_stream_sync(p->stream[g]);
    }
    for (g = 0; g < p->ndev; ++g) {
        if (cur >= 0) fa_hip_set_device(p->devs[g]);
        if (p->rows[g]) fftw_destroy_plan(p->rows[g]);
        if (p->cols[g]) fftw_destroy_plan(p->cols[g]);
        if (p->W[g]) fa_hip_free(p->W[g]);
        if (p->ev_rows[g]) fa_hip_event_destroy(p->ev_rows[g]);
        if (p->ev_x1[g]) fa_hip_event_destroy(p->ev_x1[g]);
        if (p->ev_cols[g]) fa_hip_event_destroy(p->ev_cols[g]);
        if (p->ev_done[g]) fa_hip_event_destroy(p->ev_done[g]);
        if (p->stream[g]) fa_hip_stream_destroy(p->stream[g]);
    }
    if (cur >= 0) fa_hip_set_device(cur);
    free(p);
}

/* fftw_mpi_plan_dft_2d / _3d (fftw3-mpi.h:141-152): in[g] / out[g] are device arrays on devs[g] holding its rows,
   local_n0(g) x n1 (x n2) complex values in row-major order; in == out (per device) is allowed.  NULL on invalid
   arguments, when a named device does not exist, or when a local plan / buffer cannot be made. */
struct fftw_amd_slab_plan_s *fftw_amd_slab_plan_dft(int rank, const long long *n, int ndev, const int *devs,
                                                    fftw_complex *const *in, fftw_complex *const *out,
                                                    int sign, unsigned flags) {
    struct fftw_amd_slab_plan_s *p;
    int g, h, ndevices = fa_hip_device_count(), saved;
    if (rank < 2 || rank > 3 || !n || ndev < 1 || ndev > FA_SLAB_MAXDEV || !in || !out) return NULL;
    if (n[0] <= 0 || n[1] <= 0 || (rank == 3 && n[2] <= 0) || (sign != FFTW_FORWARD && sign != FFTW_BACKWARD)) return NULL;
    if (n[0] > 0x7fffffffLL || n[1] > 0x7fffffffLL || (rank == 3 && (n[2] > 0x7fffffffLL || n[1] * n[2] > 0x7fffffffLL))) return NULL;
    p = (struct fftw_amd_slab_plan_s *)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->ndev = ndev;
    p->n0 = n[0]; p->n1 = n[1]; p->inner = rank == 3 ? n[2] : 1; p->R = p->n1 * p->inner;
    for (g = 0; g < ndev; ++g) {
        p->devs[g] = devs ? devs[g] : g;
        if (ndevices > 0 && (p->devs[g] < 0 || p->devs[g] >= ndevices)) {
            fprintf(stderr, "fftw3_amd: slab plan names device %d, but only %d are visible\n", p->devs[g], ndevices);
            free(p);
            return NULL;
        }
        slab_block(p->n0, ndev, g, &p->lo0[g], &p->ln0[g]);
        slab_block(p->n1, ndev, g, &p->lo1[g], &p->ln1[g]);
        p->in[g] = in[g]; p->out[g] = out[g];
        if (p->ln0[g] > 0 && (!in[g] || !out[g])) { free(p); return NULL; }
    }
    saved = ndevices > 0 ? fa_hip_get_device() : -1;
    for (g = 0; g < ndev; ++g) {
        const long long w = p->ln1[g] * p->inner;
        int nn[2];
        if (saved >= 0) fa_hip_set_device(p->devs[g]);
        if (saved >= 0) {
            p->stream[g] = fa_hip_stream_create();
            p->ev_rows[g] = fa_hip_event_create();
            p->ev_x1[g] = fa_hip_event_create();
            p->ev_cols[g] = fa_hip_event_create();
            p->ev_done[g] = fa_hip_event_create();
            for (h = 0; h < ndev; ++h) fa_hip_enable_peer(p->devs[g], p->devs[h]);
        }
        if (p->ln0[g] > 0) {
            /* step 1: the trailing dimension(s) of every local row */
            nn[0] = (int)p->n1; nn[1] = (int)p->inner;
            p->rows[g] = fftw_plan_many_dft(rank - 1, nn, (int)p->ln0[g], in[g], NULL, 1, (int)p->R, out[g], NULL, 1, (int)p->R, sign, flags);
            if (!p->rows[g]) goto fail;
            if (saved >= 0) fftw_amd_plan_set_stream(p->rows[g], p->stream[g]);
        }
        if (w > 0) {
            /* step 3: length-n0 transforms down the column block [n0][w], in place */
            if (saved >= 0) {
                p->W[g] = (fftw_complex *)fa_hip_malloc((size_t)p->n0 * (size_t)w * sizeof(fftw_complex));
                if (!p->W[g]) goto fail;
            }
            nn[0] = (int)p->n0;
            {
                /* without a device (CPU test tier: plan inspection only) the plan is made on a placeholder address */
                static fftw_complex placeholder[1];
                fftw_complex *wp = p->W[g] ? p->W[g] : placeholder;
                p->cols[g] = fftw_plan_many_dft(1, nn, (int)w, wp, NULL, (int)w, 1, wp, NULL, (int)w, 1, sign, flags);
            }
            if (!p->cols[g]) goto fail;
            if (saved >= 0) fftw_amd_plan_set_stream(p->cols[g], p->stream[g]);
        }
    }
    if (saved >= 0) fa_hip_set_device(saved);
    return p;
fail:
    if (saved >= 0) fa_hip_set_device(saved);
    fftw_amd_destroy_slab_plan(p);
    return NULL;
}

/* fftw_mpi_execute_dft on the plan's own arrays: enqueues everything and returns (fftw_amd_slab_sync waits) */
void fftw_amd_slab_execute(struct fftw_amd_slab_plan_s *p) {
    int g, r, saved;
    if (!p) return;
    if (fa_hip_device_count() <= 0) {
        fprintf(stderr, "fftw3_amd: no HIP device available: a slab plan cannot execute (no CPU fallback)\n");
        abort();
    }
    saved = fa_hip_get_device();
    /* 1: rows */
    for (g = 0; g < p->ndev; ++g) {
        fa_hip_set_device(p->devs[g]);
        if (p->rows[g]) fftw_execute(p->rows[g]);
        fa_hip_event_record(p->ev_rows[g], p->stream[g]);
    }
    /* 2: column block r of every device's rows -> W[r]; on the receiver's stream, behind the sender's rows */
    for (r = 0; r < p->ndev; ++r) {
        const long long w = p->ln1[r] * p->inner;
        fa_hip_set_device(p->devs[r]);
        /* W[r] is rewritten: the previous execution's exchange 4 must have read it on every device */
        for (g = 0; g < p->ndev && p->ran; ++g) fa_hip_stream_wait_event(p->stream[r], p->ev_done[g]);
        for (g = 0; g < p->ndev && w > 0; ++g) {
            if (p->ln0[g] <= 0) continue;
            fa_hip_stream_wait_event(p->stream[r], p->ev_rows[g]);
            fa_hip_memcpy2d_peer(p->W[r] + p->lo0[g] * w, (size_t)w * sizeof(fftw_complex),
                                 p->out[g] + p->lo1[r] * p->inner, (size_t)p->R * sizeof(fftw_complex),
                                 (size_t)w * sizeof(fftw_complex), (size_t)p->ln0[g], p->stream[r]);
        }
        fa_hip_event_record(p->ev_x1[r], p->stream[r]);
        /* 3: columns */
        if (p->cols[r]) fftw_execute(p->cols[r]);
        fa_hip_event_record(p->ev_cols[r], p->stream[r]);
    }
    /* 4: rows of device g from every column block -> out[g]; out[g] must no longer be read by exchange 2 */
    for (g = 0; g < p->ndev; ++g) {
        fa_hip_set_device(p->devs[g]);
        for (r = 0; r < p->ndev; ++r) fa_hip_stream_wait_event(p->stream[g], p->ev_x1[r]);
        for (r = 0; r < p->ndev && p->ln0[g] > 0; ++r) {
            const long long w = p->ln1[r] * p->inner;
            if (w <= 0) continue;
            fa_hip_stream_wait_event(p->stream[g], p->ev_cols[r]);
            fa_hip_memcpy2d_peer(p->out[g] + p->lo1[r] * p->inner, (size_t)p->R * sizeof(fftw_complex),
                                 p->W[r] + p->lo0[g] * w, (size_t)w * sizeof(fftw_complex),
                                 (size_t)w * sizeof(fftw_complex), (size_t)p->ln0[g], p->stream[g]);
        }
        fa_hip_event_record(p->ev_done[g], p->stream[g]);
    }
    p->ran = 1;
    fa_hip_set_device(saved);
}

void fftw_amd_slab_sync(struct fftw_amd_slab_plan_s *p) {
    int g, saved;
    if (!p || fa_hip_device_count() <= 0) return;
    saved = fa_hip_get_device();
    for (g = 0; g < p->ndev; ++g) {
        fa_hip_set_device(p->devs[g]);
        fa_hip_stream_sync(p->stream[g]);
    }
    fa_hip_set_device(saved);
}

int fftw_amd_slab_num_devices(const fftw_amd_slab_plan p) { return p ? p->ndev : 0; }
fftw_plan fftw_amd_slab_local_plan(const fftw_amd_slab_plan p, int g, int which) {
    if (!p || g < 0 || g >= p->ndev) return NULL;
    return which ? p->cols[g] : p->rows[g];
}
