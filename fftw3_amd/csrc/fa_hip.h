/*
 * fa_hip.h -- the thin C ABI between the C host planner (api.c / planner.c)
 * and the HIP translation unit (kernels.hip).  Plain C types only.
 */
#ifndef FA_HIP_H
#define FA_HIP_H

#include <stddef.h>
#include "fftw3_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

int   fa_hip_device_count(void);
/* sequences per tile of the two-stage register kernel for length L, 0 if there is none */
int   fa_hip_rr_tile(int L);
int   fa_hip_r3t_tile(int L);  /* sequences per tile of the strided three-stage kernel, 0: none */
int   fa_hip_r3tw_rdec(int L); /* 1: the length has the real-decimated rows form (FFTW_AMD_F_REAL_DEC) */
int   fa_hip_r2c_rows_tile(int L);  /* rows per tile of the fused real-rows kernel for half length L, 0: none */
int   fa_hip_r2c_rows1_tile(int L);  /* ... of the one-stage real-rows kernel (dense rows, half length 2 ... 32), 0: none */
int   fa_hip_r2c_rows2m_tile(int L); /* ... of the mixed-radix two-stage lengths (plain r2c / c2r, half length 72 ... 648), 0: none */
int   fa_hip_r2c_rows3_tile(int L); /* ... of its three-stage form (plain r2c / c2r, half length 2048 ... 8192), 0: none */
int   fa_hip_r3_tile(int L);   /* rows per tile of the three-stage rows kernel for length L, 0: none */
int   fa_hip_blue_nb(int need);  /* smallest padded length >= need of the one-kernel Bluestein, 0: none */
int   fa_hip_blue_tile(int nb);  /* its rows per tile */
int   fa_hip_r1_tile(int L);   /* rows per tile of the one-stage rows kernel (L = 2 ... 32), 0: none */
void *fa_hip_malloc(size_t nbytes);
void  fa_hip_free(void *p);
void *fa_hip_host_malloc(size_t nbytes);  /* pinned; NULL when no device runtime */
int   fa_hip_host_free(void *p);          /* 0 when p was not a pinned allocation */
/* 1: device-accessible pointer (hipMalloc / managed / registered), 0: plain host */
int   fa_hip_is_device_ptr(const void *p);
int   fa_hip_ptr_device(const void *p);   /* owning device of a device allocation, -1 otherwise */
void  fa_hip_memcpy_h2d(void *dst, const void *src, size_t nbytes, void *stream);
void  fa_hip_memcpy_d2h(void *dst, const void *src, size_t nbytes, void *stream);
void  fa_hip_memset(void *dst, int v, size_t nbytes, void *stream);
void  fa_hip_stream_sync(void *stream);
void *fa_hip_event_create(void);
void  fa_hip_event_record(void *ev, void *stream);
float fa_hip_event_elapsed_ms(void *start, void *stop);   /* both must have completed */
void  fa_hip_event_destroy(void *ev);
void *fa_hip_stream_create(void);                   /* non-blocking stream */
void  fa_hip_stream_destroy(void *stream);
void  fa_hip_stream_wait_event(void *stream, void *ev);
int   fa_hip_get_device(void);
void  fa_hip_set_device(int dev);                   /* of the calling host thread */
int   fa_hip_enable_peer(int dev, int peer);        /* 0: dev may address peer's memory */
void  fa_hip_memcpy_peer(void *dst, int dst_dev, const void *src, int src_dev, size_t nbytes, void *stream);
/* 2-D device-to-device copy (rows of `width` bytes, `height` of them), source and destination possibly on different
   devices with peer access; asynchronous on `stream` (a stream of the current device) */
void  fa_hip_memcpy2d_peer(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, void *stream);

/* Launch one step.  bufs[i] is the device base pointer of buffer id i, tables[i]
   the device pointer of table id i.  (chunk_start, chunk_n) select the slice
   of the batch loop when desc->batch_dim >= 0.  Returns 0 on success. */
int fa_hip_launch_step(const fftw_amd_step_desc *desc, double *const *bufs,
                       void *const *tables, long long chunk_start, long long chunk_n,
                       void *stream);

/* One launch holding pass 2 (d_second) of one chunk and pass 1 (d_first) of the next one of the
   batched 1024 x 1024 plan; either half may be empty (cn == 0).  1: the steps are not that pair. */
int fa_hip_launch_pair1024(const fftw_amd_step_desc *d_second, double *const *bufs_second,
                           long long cs2, long long cn2,
                           const fftw_amd_step_desc *d_first, double *const *bufs_first,
                           long long cs1, long long cn1, void *const *tables, void *stream);

#ifdef __cplusplus
}
#endif
#endif
