/*
 * fftw3.h -- public C interface of the MI355X-native FFT executor.
 *
 * This header is the drop-in boundary: it declares, with identical names,
 * argument order and meaning, the double-precision entry points that the
 * reference declares in fftw/fftw3.h (reference lines cited per group below),
 * so that an existing FFTW3 caller re-links against libfftw3_amd.so unchanged.
 * It is written from scratch for this project; it is not a copy of the
 * reference header (no precision-mangling macro layer, double precision only).
 *
 * Type names: the reference fork spells the element types FFTW_COMPLEX /
 * FFTW_REAL_TYPE (reference fftw/fftw3.h:65-82) while stock FFTW3 callers use
 * fftw_complex.  Both spellings are provided.
 */
#ifndef FFTW3_AMD_FFTW3_H
#define FFTW3_AMD_FFTW3_H

#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef double FFTW_REAL_TYPE;
typedef double FFTW_COMPLEX[2];
typedef double fftw_complex[2];

/* opaque plan handle (reference fftw/fftw3.h:133) */
typedef struct fftw_plan_s *fftw_plan;

/* guru dimension descriptors (reference fftw/fftw3.h:112-125) */
typedef struct fftw_iodim_s   { int n, is, os; }       fftw_iodim;
typedef struct fftw_iodim64_s { ptrdiff_t n, is, os; } fftw_iodim64;

/* r2r kinds (reference fftw/fftw3.h:84-90); all eleven are planned and executed on the
   GPU by fftw_plan_r2r_* / fftw_plan_many_r2r / fftw_plan_guru(64)_r2r. */
typedef enum {
    FFTW_R2HC = 0, FFTW_HC2R = 1, FFTW_DHT = 2,
    FFTW_REDFT00 = 3, FFTW_REDFT01 = 4, FFTW_REDFT10 = 5, FFTW_REDFT11 = 6,
    FFTW_RODFT00 = 7, FFTW_RODFT01 = 8, FFTW_RODFT10 = 9, FFTW_RODFT11 = 10
} fftw_r2r_kind;

typedef void (*fftw_write_char_func)(char c, void *);
typedef int  (*fftw_read_char_func)(void *);

/* transform direction and planner flags (reference fftw/fftw3.h:469-499) */
#define FFTW_FORWARD  (-1)
#define FFTW_BACKWARD (+1)
#define FFTW_NO_TIMELIMIT (-1.0)

#define FFTW_MEASURE          (0U)
#define FFTW_DESTROY_INPUT    (1U << 0)
#define FFTW_UNALIGNED        (1U << 1)
#define FFTW_CONSERVE_MEMORY  (1U << 2)
#define FFTW_EXHAUSTIVE       (1U << 3)
#define FFTW_PRESERVE_INPUT   (1U << 4)
#define FFTW_PATIENT          (1U << 5)
#define FFTW_ESTIMATE         (1U << 6)
#define FFTW_WISDOM_ONLY      (1U << 21)
#define FFTW_ESTIMATE_PATIENT (1U << 7)
#define FFTW_BELIEVE_PCOST    (1U << 8)
#define FFTW_NO_DFT_R2HC      (1U << 9)
#define FFTW_NO_NONTHREADED   (1U << 10)
#define FFTW_NO_BUFFERING     (1U << 11)
#define FFTW_NO_INDIRECT_OP   (1U << 12)
#define FFTW_ALLOW_LARGE_GENERIC (1U << 13)
#define FFTW_NO_RANK_SPLITS   (1U << 14)
#define FFTW_NO_VRANK_SPLITS  (1U << 15)
#define FFTW_NO_VRECURSE      (1U << 16)
#define FFTW_NO_SIMD          (1U << 17)
#define FFTW_NO_SLOW          (1U << 18)
#define FFTW_NO_FIXED_RADIX_LARGE_N (1U << 19)
#define FFTW_ALLOW_PRUNING    (1U << 20)

/* ---- execution (reference fftw/fftw3.h:144, 202-206, 317-326) ---- */
void fftw_execute(const fftw_plan p);
void fftw_execute_dft(const fftw_plan p, fftw_complex *in, fftw_complex *out);
void fftw_execute_split_dft(const fftw_plan p, double *ri, double *ii, double *ro, double *io);
void fftw_execute_dft_r2c(const fftw_plan p, double *in, fftw_complex *out);
void fftw_execute_dft_c2r(const fftw_plan p, fftw_complex *in, double *out);
void fftw_execute_split_dft_r2c(const fftw_plan p, double *in, double *ro, double *io);
void fftw_execute_split_dft_c2r(const fftw_plan p, double *ri, double *ii, double *out);
void fftw_execute_r2r(const fftw_plan p, double *in, double *out);

/* ---- complex DFT planners (reference fftw/fftw3.h:147-200) ---- */
fftw_plan fftw_plan_dft(int rank, const int *n, fftw_complex *in, fftw_complex *out,
                        int sign, unsigned flags);
fftw_plan fftw_plan_dft_1d(int n, fftw_complex *in, fftw_complex *out, int sign, unsigned flags);
fftw_plan fftw_plan_dft_2d(int n0, int n1, fftw_complex *in, fftw_complex *out,
                           int sign, unsigned flags);
fftw_plan fftw_plan_dft_3d(int n0, int n1, int n2, fftw_complex *in, fftw_complex *out,
                           int sign, unsigned flags);
fftw_plan fftw_plan_many_dft(int rank, const int *n, int howmany,
                             fftw_complex *in, const int *inembed, int istride, int idist,
                             fftw_complex *out, const int *onembed, int ostride, int odist,
                             int sign, unsigned flags);
fftw_plan fftw_plan_guru_dft(int rank, const fftw_iodim *dims,
                             int howmany_rank, const fftw_iodim *howmany_dims,
                             fftw_complex *in, fftw_complex *out, int sign, unsigned flags);
fftw_plan fftw_plan_guru_split_dft(int rank, const fftw_iodim *dims,
                                   int howmany_rank, const fftw_iodim *howmany_dims,
                                   double *ri, double *ii, double *ro, double *io, unsigned flags);
fftw_plan fftw_plan_guru64_dft(int rank, const fftw_iodim64 *dims,
                               int howmany_rank, const fftw_iodim64 *howmany_dims,
                               fftw_complex *in, fftw_complex *out, int sign, unsigned flags);
fftw_plan fftw_plan_guru64_split_dft(int rank, const fftw_iodim64 *dims,
                                     int howmany_rank, const fftw_iodim64 *howmany_dims,
                                     double *ri, double *ii, double *ro, double *io, unsigned flags);

/* ---- real-data DFT planners (reference fftw/fftw3.h:209-315) ---- */
fftw_plan fftw_plan_many_dft_r2c(int rank, const int *n, int howmany,
                                 double *in, const int *inembed, int istride, int idist,
                                 fftw_complex *out, const int *onembed, int ostride, int odist,
                                 unsigned flags);
fftw_plan fftw_plan_dft_r2c(int rank, const int *n, double *in, fftw_complex *out, unsigned flags);
fftw_plan fftw_plan_dft_r2c_1d(int n, double *in, fftw_complex *out, unsigned flags);
fftw_plan fftw_plan_dft_r2c_2d(int n0, int n1, double *in, fftw_complex *out, unsigned flags);
fftw_plan fftw_plan_dft_r2c_3d(int n0, int n1, int n2, double *in, fftw_complex *out, unsigned flags);
fftw_plan fftw_plan_many_dft_c2r(int rank, const int *n, int howmany,
                                 fftw_complex *in, const int *inembed, int istride, int idist,
                                 double *out, const int *onembed, int ostride, int odist,
                                 unsigned flags);
fftw_plan fftw_plan_dft_c2r(int rank, const int *n, fftw_complex *in, double *out, unsigned flags);
fftw_plan fftw_plan_dft_c2r_1d(int n, fftw_complex *in, double *out, unsigned flags);
fftw_plan fftw_plan_dft_c2r_2d(int n0, int n1, fftw_complex *in, double *out, unsigned flags);
fftw_plan fftw_plan_dft_c2r_3d(int n0, int n1, int n2, fftw_complex *in, double *out, unsigned flags);
fftw_plan fftw_plan_guru_dft_r2c(int rank, const fftw_iodim *dims,
                                 int howmany_rank, const fftw_iodim *howmany_dims,
                                 double *in, fftw_complex *out, unsigned flags);
fftw_plan fftw_plan_guru_dft_c2r(int rank, const fftw_iodim *dims,
                                 int howmany_rank, const fftw_iodim *howmany_dims,
                                 fftw_complex *in, double *out, unsigned flags);
fftw_plan fftw_plan_guru_split_dft_r2c(int rank, const fftw_iodim *dims,
                                       int howmany_rank, const fftw_iodim *howmany_dims,
                                       double *in, double *ro, double *io, unsigned flags);
fftw_plan fftw_plan_guru_split_dft_c2r(int rank, const fftw_iodim *dims,
                                       int howmany_rank, const fftw_iodim *howmany_dims,
                                       double *ri, double *ii, double *out, unsigned flags);
fftw_plan fftw_plan_guru64_dft_r2c(int rank, const fftw_iodim64 *dims,
                                   int howmany_rank, const fftw_iodim64 *howmany_dims,
                                   double *in, fftw_complex *out, unsigned flags);
fftw_plan fftw_plan_guru64_dft_c2r(int rank, const fftw_iodim64 *dims,
                                   int howmany_rank, const fftw_iodim64 *howmany_dims,
                                   fftw_complex *in, double *out, unsigned flags);
fftw_plan fftw_plan_guru64_split_dft_r2c(int rank, const fftw_iodim64 *dims,
                                         int howmany_rank, const fftw_iodim64 *howmany_dims,
                                         double *in, double *ro, double *io, unsigned flags);
fftw_plan fftw_plan_guru64_split_dft_c2r(int rank, const fftw_iodim64 *dims,
                                         int howmany_rank, const fftw_iodim64 *howmany_dims,
                                         double *ri, double *ii, double *out, unsigned flags);

/* ---- r2r planners (reference fftw/fftw3.h:328-372) ---- */
fftw_plan fftw_plan_many_r2r(int rank, const int *n, int howmany,
                             double *in, const int *inembed, int istride, int idist,
                             double *out, const int *onembed, int ostride, int odist,
                             const fftw_r2r_kind *kind, unsigned flags);
fftw_plan fftw_plan_r2r(int rank, const int *n, double *in, double *out,
                        const fftw_r2r_kind *kind, unsigned flags);
fftw_plan fftw_plan_r2r_1d(int n, double *in, double *out, fftw_r2r_kind kind, unsigned flags);
fftw_plan fftw_plan_r2r_2d(int n0, int n1, double *in, double *out,
                           fftw_r2r_kind kind0, fftw_r2r_kind kind1, unsigned flags);
fftw_plan fftw_plan_r2r_3d(int n0, int n1, int n2, double *in, double *out,
                           fftw_r2r_kind kind0, fftw_r2r_kind kind1, fftw_r2r_kind kind2,
                           unsigned flags);
fftw_plan fftw_plan_guru_r2r(int rank, const fftw_iodim *dims,
                             int howmany_rank, const fftw_iodim *howmany_dims,
                             double *in, double *out, const fftw_r2r_kind *kind, unsigned flags);
fftw_plan fftw_plan_guru64_r2r(int rank, const fftw_iodim64 *dims,
                               int howmany_rank, const fftw_iodim64 *howmany_dims,
                               double *in, double *out, const fftw_r2r_kind *kind, unsigned flags);

/* ---- plan lifetime, planner state (reference fftw/fftw3.h:376-400) ---- */
void fftw_destroy_plan(fftw_plan p);
void fftw_forget_wisdom(void);
void fftw_cleanup(void);
void fftw_set_timelimit(double t);
void fftw_plan_with_nthreads(int nthreads);
int  fftw_init_threads(void);
void fftw_cleanup_threads(void);
void fftw_make_planner_thread_safe(void);

/* ---- wisdom (reference fftw/fftw3.h:402-426) ---- */
int   fftw_export_wisdom_to_filename(const char *filename);
void  fftw_export_wisdom_to_file(FILE *output_file);
char *fftw_export_wisdom_to_string(void);
void  fftw_export_wisdom(fftw_write_char_func write_char, void *data);
int   fftw_import_system_wisdom(void);
int   fftw_import_wisdom_from_filename(const char *filename);
int   fftw_import_wisdom_from_file(FILE *input_file);
int   fftw_import_wisdom_from_string(const char *input_string);
int   fftw_import_wisdom(fftw_read_char_func read_char, void *data);

/* ---- plan introspection (reference fftw/fftw3.h:428-434, 449-456) ---- */
void   fftw_fprint_plan(const fftw_plan p, FILE *output_file);
void   fftw_print_plan(const fftw_plan p);
char  *fftw_sprint_plan(const fftw_plan p);
void   fftw_flops(const fftw_plan p, double *add, double *mul, double *fmas);
double fftw_estimate_cost(const fftw_plan p);
double fftw_cost(const fftw_plan p);

/* ---- memory (reference fftw/fftw3.h:437-446, 459) ---- */
void   *fftw_malloc(size_t n);
double *fftw_alloc_real(size_t n);
fftw_complex *fftw_alloc_complex(size_t n);
void    fftw_free(void *p);
int     fftw_alignment_of(double *p);

/* ---- version strings (reference fftw/fftw3.h:461-463) ---- */
extern const char fftw_version[];
extern const char fftw_cc[];
extern const char fftw_codelet_optim[];

#ifdef __cplusplus
}
#endif
#endif /* FFTW3_AMD_FFTW3_H */
