/*
 * fftw_oracle.c -- CPU restatement of the reference's codelet-path algorithm.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under fftw3_amd/ (the product) includes,
 * links or calls this file; it is the checker for tests/, for
 * __graft_entry__.smoke() and for bench.py's cpu_baseline leg.
 *
 * PINNING STATUS: "parity unpinned" in the strict sense of the project rules -- no output of the
 * reference library itself (neither a build of it nor vectors stored in its tree) anchors this
 * file; what anchors it is listed below.
 *
 * PINNING: the reference itself could not be built under the rules of this
 * project (fftw/fftw_api.h:57 includes a config.h that only the reference's
 * CMake configure step generates, and that build system may not be run), and
 * the reference ships no stored golden vectors.  The
 * oracle is therefore pinned by the known-answer and self-checking tests the
 * reference's own verifier holds for this path -- impulse / constant input,
 * linearity, time- and frequency-shift at relative tolerance 1e-10
 * (fftw/libbench2/verify-lib.c:284-414, bench-main.c:70) -- restated in
 * tests/verifier.py, plus closed-form single-tone answers and an 80-bit
 * long-double direct DFT (tests/test_oracle_pin.py, tests/golden/).
 *
 * What is restated (all double precision, plain C loops):
 *   oracle_cexp        real_cexp octant reduction      fftw/fftw_api.c:18850-18892
 *   twiddle cache      mktwiddle / TW_FULL rows        fftw/fftw_api.c:19134-19231
 *   dft_rec            ct_apply_dit: child DFTs on the r decimated inputs into
 *                      r blocks of m, then the twiddle butterfly across the
 *                      blocks (dftw_direct_apply)      fftw/fftw_api.c:2078-2202, 2315-2324
 *   butterfly_small    the n1/t1 codelets' arithmetic: x_i * conj(W_i), then a
 *                      size-r DFT                      fftw/dft_scalar/codelets/t1_4.c:125-191
 *   dft_generic        O(n^2) odd-prime DFT            fftw/fftw_api.c:3390-3448
 *   dft_rader          Rader, smallest generator, conj trick
 *                                                      fftw/fftw_api.c:4139-4261, 15812-15828
 *   dft_bluestein      chirp-z with k^2 mod 2n         fftw/fftw_api.c:1598-1688
 *   r2c / c2r          half-length complex DFT + hc2cfdft/hc2cbdft untangle
 *                                                      fftw/fftw_api.c:5579-5590, 5831-5845
 *   *_many addressing  fftw_plan_many_dft tensors      fftw/fftw_api.c:642-666, 774-788, 842-861
 * The reference's planner picks radices by a cost model (which radix is used
 * changes only rounding); this restatement uses radix 8/4/2 for powers of two
 * and the smallest odd prime otherwise.  Odd-length r2c goes through a complex
 * DFT of the real sequence (the reference uses its halfcomplex solvers there).
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef double complex cx;
typedef long long i64;

/* ---- (cos, sin)(2 pi m / n) with the argument folded into [0, pi/4] ---- */
void oracle_cexp(i64 m, i64 n, double *out) {
    const double K2PI = 6.2831853071795864769252867665590057683943388;
    i64 quarter_n = n;
    unsigned octant = 0;
    double theta, c, s, t;
    m %= n;
    n += n; n += n;
    m += m; m += m;
    if (m < 0) m += n;
    if (m > n - m) { m = n - m; octant |= 4; }
    if (m - quarter_n > 0) { m = m - quarter_n; octant |= 2; }
    if (m > quarter_n - m) { m = quarter_n - m; octant |= 1; }
    theta = (K2PI * (double)m) / (double)n;
    c = cos(theta);
    s = sin(theta);
    if (octant & 1) { t = c; c = s; s = t; }
    if (octant & 2) { t = c; c = -s; s = t; }
    if (octant & 4) { s = -s; }
    out[0] = c;
    out[1] = s;
}

/* ---- twiddle cache: W_n[k] = cos + i sin (2 pi k / n), k in [0, n) ---- */
#define MAXTW 256
static struct { i64 n; cx *w; } twcache[MAXTW];
static int ntw = 0;

static const cx *twiddles(i64 n) {
    int i;
    i64 k;
    cx *w;
    for (i = 0; i < ntw; ++i) if (twcache[i].n == n) return twcache[i].w;
    w = (cx *)malloc(sizeof(cx) * (size_t)n);
    for (k = 0; k < n; ++k) {
        double cs[2];
        oracle_cexp(k, n, cs);
        w[k] = cs[0] + I * cs[1];
    }
    if (ntw == MAXTW) abort();   /* one transform needs far fewer tables; see cache_gc */
    twcache[ntw].n = n;
    twcache[ntw].w = w;
    ++ntw;
    return w;
}

void oracle_forget(void) {
    int i;
    for (i = 0; i < ntw; ++i) free(twcache[i].w);
    ntw = 0;
}

/* tables are only dropped between transforms, never while one is running */
static void cache_gc(void) { if (ntw > MAXTW / 2) oracle_forget(); }

/* ---- number theory ---- */
static i64 mulmod(i64 a, i64 b, i64 p) { return (i64)(((unsigned __int128)a * (unsigned __int128)b) % (unsigned __int128)p); }
static i64 powmod(i64 b, i64 e, i64 p) {
    i64 r = 1;
    b %= p;
    while (e) { if (e & 1) r = mulmod(r, b, p); b = mulmod(b, b, p); e >>= 1; }
    return r;
}
static i64 first_divisor(i64 n) {
    i64 i;
    if (n <= 1) return n;
    if (n % 2 == 0) return 2;
    for (i = 3; i * i <= n; i += 2) if (n % i == 0) return i;
    return n;
}
static int is_prime(i64 n) { return n > 1 && first_divisor(n) == n; }
static i64 find_generator(i64 p) {
    i64 pf[64], m = p - 1, g, d;
    int k = 0, i;
    if (p == 2) return 1;
    for (d = 2; d * d <= m; ++d) if (m % d == 0) { pf[k++] = d; while (m % d == 0) m /= d; }
    if (m > 1) pf[k++] = m;
    for (g = 2;; ++g) {
        int ok = 1;
        for (i = 0; i < k; ++i) if (powmod(g, (p - 1) / pf[i], p) == 1) { ok = 0; break; }
        if (ok) return g;
    }
}
static int smooth235(i64 n) {
    while (n % 2 == 0) n /= 2;
    while (n % 3 == 0) n /= 3;
    while (n % 5 == 0) n /= 5;
    return n == 1;
}

static void dft_rec(i64 n, const cx *in, i64 is, cx *out, i64 os);

/* forward size-r DFT of t[0..r) in place, r small */
static void butterfly_small(int r, cx *t) {
    if (r == 2) {
        cx a = t[0];
        t[0] = a + t[1];
        t[1] = a - t[1];
    } else if (r == 4) {
        cx a = t[0] + t[2], b = t[0] - t[2], c = t[1] + t[3], d = (t[1] - t[3]) * (-I);
        t[0] = a + c; t[2] = a - c; t[1] = b + d; t[3] = b - d;
    } else if (r == 8) {
        cx e[4] = { t[0], t[2], t[4], t[6] }, o[4] = { t[1], t[3], t[5], t[7] };
        const cx *w = twiddles(8);
        int k;
        butterfly_small(4, e);
        butterfly_small(4, o);
        for (k = 0; k < 4; ++k) {
            cx ow = o[k] * conj(w[k]);
            t[k] = e[k] + ow;
            t[k + 4] = e[k] - ow;
        }
    } else {
        /* odd prime: X[q] = sum_j x[j] conj(W_r[jq mod r]) */
        const cx *w = twiddles(r);
        cx x[16], acc;
        int q, j;
        for (j = 0; j < r; ++j) x[j] = t[j];
        for (q = 0; q < r; ++q) {
            acc = x[0];
            for (j = 1; j < r; ++j) acc += x[j] * conj(w[(j * q) % r]);
            t[q] = acc;
        }
    }
}

/* O(n^2) DFT for odd primes below the Rader/Bluestein thresholds: folds
   x[j] +- x[n-j] and forms (n-1)/2 real dot products per output pair */
static void dft_generic(i64 n, const cx *in, i64 is, cx *out, i64 os) {
    const cx *w = twiddles(n);
    i64 h = (n - 1) / 2, j, q;
    cx *a = (cx *)malloc(sizeof(cx) * (size_t)(h + 1)), *b = (cx *)malloc(sizeof(cx) * (size_t)(h + 1));
    cx x0 = in[0], sum = in[0];
    for (j = 1; j <= h; ++j) {
        a[j] = in[j * is] + in[(n - j) * is];
        b[j] = in[j * is] - in[(n - j) * is];
        sum += a[j];
    }
    out[0] = sum;
    for (q = 1; q <= h; ++q) {
        double ur = creal(x0), ui = cimag(x0), vr = 0, vi = 0;
        for (j = 1; j <= h; ++j) {
            cx wq = w[(j * q) % n];
            ur += creal(wq) * creal(a[j]); ui += creal(wq) * cimag(a[j]);
            vr += cimag(wq) * creal(b[j]); vi += cimag(wq) * cimag(b[j]);
        }
        out[q * os] = (ur + vi) + I * (ui - vr);
        out[(n - q) * os] = (ur - vi) + I * (ui + vr);
    }
    free(a);
    free(b);
}

/* Rader: a[k] = x[g^k]; A = DFT(a); B = conj(A * Omega), B[0] += conj(x0);
   c = DFT(B); Y[g^-k] = conj(c[k]); Y[0] = x0 + A[0] */
static void dft_rader(i64 p, const cx *in, i64 is, cx *out, i64 os) {
    i64 m = p - 1, g = find_generator(p), ginv = powmod(g, p - 2, p), k, gp;
    const cx *wp = twiddles(p);
    cx *a = (cx *)malloc(sizeof(cx) * (size_t)m), *A = (cx *)malloc(sizeof(cx) * (size_t)m);
    cx *om = (cx *)malloc(sizeof(cx) * (size_t)m), *Om = (cx *)malloc(sizeof(cx) * (size_t)m);
    cx x0 = in[0];
    for (k = 0, gp = 1; k < m; ++k, gp = mulmod(gp, ginv, p)) om[k] = conj(wp[gp]) / (double)m;
    dft_rec(m, om, 1, Om, 1);
    for (k = 0, gp = 1; k < m; ++k, gp = mulmod(gp, g, p)) a[k] = in[gp * is];
    dft_rec(m, a, 1, A, 1);
    out[0] = x0 + A[0];
    for (k = 0; k < m; ++k) a[k] = conj(A[k] * Om[k]);
    a[0] += conj(x0);
    dft_rec(m, a, 1, A, 1);
    for (k = 0, gp = 1; k < m; ++k, gp = mulmod(gp, ginv, p)) out[gp * os] = conj(A[k]);
    free(a); free(A); free(om); free(Om);
}

/* Bluestein: w[k] = exp(i pi k^2 / n) with k^2 taken mod 2n */
static void dft_bluestein(i64 n, const cx *in, i64 is, cx *out, i64 os) {
    i64 nb = 2 * n - 1, k;
    const cx *w2 = twiddles(2 * n);
    cx *w, *b, *B, *W;
    while (!smooth235(nb)) ++nb;
    w = (cx *)malloc(sizeof(cx) * (size_t)n);
    b = (cx *)calloc((size_t)nb, sizeof(cx));
    B = (cx *)malloc(sizeof(cx) * (size_t)nb);
    W = (cx *)malloc(sizeof(cx) * (size_t)nb);
    for (k = 0; k < n; ++k) w[k] = w2[mulmod(k, k, 2 * n)];
    b[0] = w[0] / (double)nb;
    for (k = 1; k < n; ++k) b[k] = b[nb - k] = w[k] / (double)nb;
    dft_rec(nb, b, 1, W, 1);
    memset(b, 0, sizeof(cx) * (size_t)nb);
    for (k = 0; k < n; ++k) b[k] = in[k * is] * conj(w[k]);
    dft_rec(nb, b, 1, B, 1);
    /* inverse DFT by the swap identity: idft(x) = swap(dft(swap(x))) */
    for (k = 0; k < nb; ++k) { cx v = B[k] * W[k]; b[k] = cimag(v) + I * creal(v); }
    dft_rec(nb, b, 1, B, 1);
    for (k = 0; k < n; ++k) { cx v = cimag(B[k]) + I * creal(B[k]); out[k * os] = v * conj(w[k]); }
    free(w); free(b); free(B); free(W);
}

static int choose_radix(i64 n) {
    i64 d;
    if (n % 8 == 0 && n > 8) return 8;
    if (n % 4 == 0 && n > 4) return 4;
    if (n % 2 == 0 && n > 2) return 2;
    d = first_divisor(n);
    return (d <= 13 && d < n) ? (int)d : 0;
}

/* forward DFT, out-of-place, in and out must not overlap */
static void dft_rec(i64 n, const cx *in, i64 is, cx *out, i64 os) {
    int r;
    i64 m, i, k;
    if (n == 1) { out[0] = in[0]; return; }
    if (n == 2 || n == 4 || n == 8 || (n <= 13 && is_prime(n))) {
        cx t[16];
        for (i = 0; i < n; ++i) t[i] = in[i * is];
        butterfly_small((int)n, t);
        for (i = 0; i < n; ++i) out[i * os] = t[i];
        return;
    }
    r = choose_radix(n);
    if (r == 0) {
        /* no small factor left: n is prime > 13 or has only large prime factors */
        i64 d = first_divisor(n);
        if (d == n) {
            if (n > 32 && smooth235(n - 1)) dft_rader(n, in, is, out, os);
            else if (n < 173) dft_generic(n, in, is, out, os);
            else dft_bluestein(n, in, is, out, os);
            return;
        }
        /* composite of large primes: Cooley-Tukey with the generic/Rader/
           Bluestein transform as the butterfly */
        {
            const cx *w = twiddles(n);
            cx *t, *u;
            m = n / d;
            for (i = 0; i < d; ++i) dft_rec(m, in + i * is, d * is, out + i * m * os, os);
            t = (cx *)malloc(sizeof(cx) * (size_t)d);
            u = (cx *)malloc(sizeof(cx) * (size_t)d);
            for (k = 0; k < m; ++k) {
                for (i = 0; i < d; ++i) t[i] = out[(i * m + k) * os] * conj(w[(i * k) % n]);
                dft_rec(d, t, 1, u, 1);
                for (i = 0; i < d; ++i) out[(k + m * i) * os] = u[i];
            }
            free(t); free(u);
            return;
        }
    }
    m = n / r;
    /* step 1: r DFTs of size m on the decimated inputs, into r blocks of m */
    for (i = 0; i < r; ++i) dft_rec(m, in + i * is, r * is, out + i * m * os, os);
    /* step 2: twiddle and size-r butterflies across the blocks, in place */
    {
        const cx *w = twiddles(n);
        cx t[16];
        for (k = 0; k < m; ++k) {
            t[0] = out[k * os];
            for (i = 1; i < r; ++i) t[i] = out[(i * m + k) * os] * conj(w[i * k]);
            butterfly_small(r, t);
            for (i = 0; i < r; ++i) out[(k + m * i) * os] = t[i];
        }
    }
}

/* forward or backward (sign = +1) by the (re,im) swap identity, A.c:14555 */
static void dft_1d(i64 n, cx *buf, i64 stride, int sign, cx *tmp_in, cx *tmp_out) {
    i64 k;
    if (sign > 0) for (k = 0; k < n; ++k) { cx v = buf[k * stride]; tmp_in[k] = cimag(v) + I * creal(v); }
    else for (k = 0; k < n; ++k) tmp_in[k] = buf[k * stride];
    dft_rec(n, tmp_in, 1, tmp_out, 1);
    if (sign > 0) for (k = 0; k < n; ++k) { cx v = tmp_out[k]; buf[k * stride] = cimag(v) + I * creal(v); }
    else for (k = 0; k < n; ++k) buf[k * stride] = tmp_out[k];
}

/* separable row-major transform of a dense rank-d array, in place */
static void dft_nd_dense(int rank, const i64 *n, cx *a, int sign) {
    i64 total = 1, maxn = 1, stride = 1;
    int d;
    cx *ti, *to;
    for (d = 0; d < rank; ++d) { total *= n[d]; if (n[d] > maxn) maxn = n[d]; }
    ti = (cx *)malloc(sizeof(cx) * (size_t)maxn);
    to = (cx *)malloc(sizeof(cx) * (size_t)maxn);
    for (d = rank - 1; d >= 0; --d) {
        i64 outer = total / (n[d] * stride), o, s;
        for (o = 0; o < outer; ++o)
            for (s = 0; s < stride; ++s)
                dft_1d(n[d], a + o * n[d] * stride + s, stride, sign, ti, to);
        stride *= n[d];
    }
    free(ti);
    free(to);
}

/* row-major offset of a flat dense index into an embedded array */
static i64 embed_off(int rank, const i64 *n, const i64 *emb, i64 flat, i64 stride) {
    i64 off = 0, mul = stride;
    int d;
    for (d = rank - 1; d >= 0; --d) {
        off += (flat % n[d]) * mul;
        flat /= n[d];
        mul *= emb[d];
    }
    return off;
}

/* fftw_plan_many_dft + fftw_execute on interleaved complex host arrays */
int oracle_dft_many(int rank, const int *n, int howmany,
                    const double *in, const int *inembed, int istride, int idist,
                    double *out, const int *onembed, int ostride, int odist, int sign) {
    i64 nn[16], ie[16], oe[16], total = 1, f;
    int d, b;
    cx *a;
    if (rank < 0 || rank > 16 || howmany < 0) return -1;
    cache_gc();
    for (d = 0; d < rank; ++d) {
        if (n[d] <= 0) return -1;
        nn[d] = n[d];
        ie[d] = inembed ? inembed[d] : n[d];
        oe[d] = onembed ? onembed[d] : n[d];
        total *= n[d];
    }
    a = (cx *)malloc(sizeof(cx) * (size_t)total);
    for (b = 0; b < howmany; ++b) {
        const double *ib = in + 2 * (i64)b * idist;
        double *ob = out + 2 * (i64)b * odist;
        for (f = 0; f < total; ++f) {
            i64 o = embed_off(rank, nn, ie, f, istride);
            a[f] = ib[2 * o] + I * ib[2 * o + 1];
        }
        dft_nd_dense(rank, nn, a, sign);
        for (f = 0; f < total; ++f) {
            i64 o = embed_off(rank, nn, oe, f, ostride);
            ob[2 * o] = creal(a[f]);
            ob[2 * o + 1] = cimag(a[f]);
        }
    }
    free(a);
    return 0;
}

/* r2c along a contiguous row of n reals -> n/2+1 complex */
static void r2c_row(i64 n, const double *x, cx *y) {
    i64 k;
    if (n % 2 == 0) {
        i64 h = n / 2;
        const cx *w = twiddles(n);
        cx *z = (cx *)malloc(sizeof(cx) * (size_t)h), *Z = (cx *)malloc(sizeof(cx) * (size_t)h);
        for (k = 0; k < h; ++k) z[k] = x[2 * k] + I * x[2 * k + 1];
        dft_rec(h, z, 1, Z, 1);
        for (k = 0; k <= h; ++k) {
            cx zk = Z[k % h], zm = conj(Z[(h - k) % h]);
            cx E = 0.5 * (zk + zm), O = -0.5 * I * (zk - zm);
            cx wk = (k == h) ? -1.0 : conj(w[k]);
            y[k] = E + wk * O;
        }
        y[0] = creal(y[0]);
        y[h] = creal(y[h]);
        free(z); free(Z);
    } else {
        cx *z = (cx *)malloc(sizeof(cx) * (size_t)n), *Z = (cx *)malloc(sizeof(cx) * (size_t)n);
        for (k = 0; k < n; ++k) z[k] = x[k];
        dft_rec(n, z, 1, Z, 1);
        for (k = 0; k <= n / 2; ++k) y[k] = Z[k];
        y[0] = creal(y[0]);
        free(z); free(Z);
    }
}

/* c2r along a row: n/2+1 complex -> n reals (unnormalised) */
static void c2r_row(i64 n, const cx *y, double *x) {
    i64 k;
    if (n % 2 == 0) {
        i64 h = n / 2;
        const cx *w = twiddles(n);
        cx *z = (cx *)malloc(sizeof(cx) * (size_t)h), *Z = (cx *)malloc(sizeof(cx) * (size_t)h);
        for (k = 0; k < h; ++k) {
            cx yk = y[k], ym = conj(y[h - k]);
            cx E, O, v;
            if (k == 0) { yk = creal(y[0]); ym = creal(y[h]); }
            E = yk + ym;
            O = (yk - ym) * w[k];
            v = E + I * O;
            z[k] = cimag(v) + I * creal(v);           /* swap: backward by a forward DFT */
        }
        dft_rec(h, z, 1, Z, 1);
        for (k = 0; k < h; ++k) { x[2 * k] = cimag(Z[k]); x[2 * k + 1] = creal(Z[k]); }
        free(z); free(Z);
    } else {
        cx *z = (cx *)malloc(sizeof(cx) * (size_t)n), *Z = (cx *)malloc(sizeof(cx) * (size_t)n);
        for (k = 0; k < n; ++k) {
            cx v = (k <= n / 2) ? y[k] : conj(y[n - k]);
            if (k == 0) v = creal(v);
            z[k] = cimag(v) + I * creal(v);
        }
        dft_rec(n, z, 1, Z, 1);
        for (k = 0; k < n; ++k) x[k] = cimag(Z[k]);
        free(z); free(Z);
    }
}

/* fftw_plan_many_dft_r2c: in = reals, out = interleaved complex.  NULL embeds
   follow fftw_rdft2_pad (A.c:774-788) for the out-of-place case; in-place
   callers pass explicit embeds. */
int oracle_r2c_many(int rank, const int *n, int howmany,
                    const double *in, const int *inembed, int istride, int idist,
                    double *out, const int *onembed, int ostride, int odist) {
    i64 nn[16], hn[16], ie[16], oe[16], total = 1, htotal = 1, rows, nl, hl, f, r;
    int d, b;
    double *x;
    cx *y;
    if (rank < 1 || rank > 16 || howmany < 0) return -1;
    cache_gc();
    for (d = 0; d < rank; ++d) {
        if (n[d] <= 0) return -1;
        nn[d] = hn[d] = n[d];
        ie[d] = inembed ? inembed[d] : n[d];
        oe[d] = onembed ? onembed[d] : n[d];
    }
    nl = n[rank - 1];
    hl = nl / 2 + 1;
    hn[rank - 1] = hl;
    if (!onembed) oe[rank - 1] = hl;
    for (d = 0; d < rank; ++d) { total *= nn[d]; htotal *= hn[d]; }
    rows = total / nl;
    x = (double *)malloc(sizeof(double) * (size_t)total);
    y = (cx *)malloc(sizeof(cx) * (size_t)htotal);
    for (b = 0; b < howmany; ++b) {
        const double *ib = in + (i64)b * idist;
        double *ob = out + 2 * (i64)b * odist;
        for (f = 0; f < total; ++f) x[f] = ib[embed_off(rank, nn, ie, f, istride)];
        for (r = 0; r < rows; ++r) r2c_row(nl, x + r * nl, y + r * hl);
        /* complex DFTs along the leading dims of the half-spectrum array */
        if (rank > 1) {
            i64 stride = hl, maxn = 1;
            cx *ti, *to;
            for (d = 0; d < rank - 1; ++d) if (nn[d] > maxn) maxn = nn[d];
            ti = (cx *)malloc(sizeof(cx) * (size_t)maxn);
            to = (cx *)malloc(sizeof(cx) * (size_t)maxn);
            for (d = rank - 2; d >= 0; --d) {
                i64 outer = htotal / (nn[d] * stride), o, s;
                for (o = 0; o < outer; ++o)
                    for (s = 0; s < stride; ++s)
                        dft_1d(nn[d], y + o * nn[d] * stride + s, stride, -1, ti, to);
                stride *= nn[d];
            }
            free(ti); free(to);
        }
        for (f = 0; f < htotal; ++f) {
            i64 o = embed_off(rank, hn, oe, f, ostride);
            ob[2 * o] = creal(y[f]);
            ob[2 * o + 1] = cimag(y[f]);
        }
    }
    free(x); free(y);
    return 0;
}

int oracle_c2r_many(int rank, const int *n, int howmany,
                    const double *in, const int *inembed, int istride, int idist,
                    double *out, const int *onembed, int ostride, int odist) {
    i64 nn[16], hn[16], ie[16], oe[16], total = 1, htotal = 1, rows, nl, hl, f, r;
    int d, b;
    double *x;
    cx *y;
    if (rank < 1 || rank > 16 || howmany < 0) return -1;
    cache_gc();
    for (d = 0; d < rank; ++d) {
        if (n[d] <= 0) return -1;
        nn[d] = hn[d] = n[d];
        ie[d] = inembed ? inembed[d] : n[d];
        oe[d] = onembed ? onembed[d] : n[d];
    }
    nl = n[rank - 1];
    hl = nl / 2 + 1;
    hn[rank - 1] = hl;
    if (!inembed) ie[rank - 1] = hl;
    for (d = 0; d < rank; ++d) { total *= nn[d]; htotal *= hn[d]; }
    rows = total / nl;
    x = (double *)malloc(sizeof(double) * (size_t)total);
    y = (cx *)malloc(sizeof(cx) * (size_t)htotal);
    for (b = 0; b < howmany; ++b) {
        const double *ib = in + 2 * (i64)b * idist;
        double *ob = out + (i64)b * odist;
        for (f = 0; f < htotal; ++f) {
            i64 o = embed_off(rank, hn, ie, f, istride);
            y[f] = ib[2 * o] + I * ib[2 * o + 1];
        }
        if (rank > 1) {
            i64 stride = hl, maxn = 1;
            cx *ti, *to;
            for (d = 0; d < rank - 1; ++d) if (nn[d] > maxn) maxn = nn[d];
            ti = (cx *)malloc(sizeof(cx) * (size_t)maxn);
            to = (cx *)malloc(sizeof(cx) * (size_t)maxn);
            for (d = rank - 2; d >= 0; --d) {
                i64 outer = htotal / (nn[d] * stride), o, s;
                for (o = 0; o < outer; ++o)
                    for (s = 0; s < stride; ++s)
                        dft_1d(nn[d], y + o * nn[d] * stride + s, stride, +1, ti, to);
                stride *= nn[d];
            }
            free(ti); free(to);
        }
        for (r = 0; r < rows; ++r) c2r_row(nl, y + r * hl, x + r * nl);
        for (f = 0; f < total; ++f) ob[embed_off(rank, nn, oe, f, ostride)] = x[f];
    }
    free(x); free(y);
    return 0;
}

/* ------------------------------------------------------------------ r2r */
/* r2r kinds, same numbering as fftw/fftw3.h:108-112 */
enum { K_R2HC = 0, K_HC2R, K_DHT, K_REDFT00, K_REDFT01, K_REDFT10, K_REDFT11,
       K_RODFT00, K_RODFT01, K_RODFT10, K_RODFT11 };

/* One row of an r2r transform, restating how the reference's own accuracy test
   defines every kind: as a DFT of logical size n0 of a sequence made real and
   even/odd by the constraint functions mkre00 / mkro00 / mkre01 / mkro01 /
   mkre10 / mkio10 / mkre11 / mkro11 (fftw/libbench2/verify-r2r.c:703-800), with
   the sample placement and the 1/2 scalings of r2r_apply (:802-925) and
   n0 = n, 2(n-1), 2(n+1), 4n or 8n (accuracy_r2r :944-955).  The size-n0 real
   DFT is r2c_row above, i.e. the pinned complex path. */
static int r2r_row(int kind, i64 n, const double *x, double *y) {
    i64 n0, j, k;
    double *e;
    cx *F;
    switch (kind) {
    case K_R2HC: case K_DHT: case K_HC2R: n0 = n; break;
    case K_REDFT00: if (n < 2) return -1; n0 = 2 * (n - 1); break;
    case K_RODFT00: n0 = 2 * (n + 1); break;
    case K_REDFT01: case K_REDFT10: case K_RODFT01: case K_RODFT10: n0 = 4 * n; break;
    case K_REDFT11: case K_RODFT11: n0 = 8 * n; break;
    default: return -1;
    }
    e = (double *)calloc((size_t)n0, sizeof(double));
    F = (cx *)malloc(sizeof(cx) * (size_t)(n0 / 2 + 1));
    if (kind == K_HC2R) {
        /* halfcomplex r0 r1 .. r(n/2) i((n+1)/2-1) .. i1 -> Hermitian spectrum, backward DFT */
        for (k = 0; k <= n / 2; ++k) {
            double im = (k > 0 && 2 * k < n) ? x[n - k] : 0.0;
            F[k] = x[k] + I * im;
        }
        c2r_row(n, F, y);
        free(e); free(F);
        return 0;
    }
    switch (kind) {
    case K_R2HC: case K_DHT:
        for (j = 0; j < n; ++j) e[j] = x[j];
        break;
    case K_REDFT00:                       /* even about 0 and n-1 */
        for (j = 0; j < n; ++j) { e[j] = x[j]; e[(n0 - j) % n0] = x[j]; }
        break;
    case K_RODFT00:                       /* odd about -1 and n */
        for (j = 0; j < n; ++j) { e[j + 1] = x[j]; e[n0 - 1 - j] = -x[j]; }
        break;
    case K_REDFT10:                       /* samples at odd points of the 4n grid, even */
        for (j = 0; j < n; ++j) { e[2 * j + 1] = x[j]; e[n0 - 2 * j - 1] = x[j]; }
        break;
    case K_RODFT10:                       /* the same, odd */
        for (j = 0; j < n; ++j) { e[2 * j + 1] = x[j]; e[n0 - 2 * j - 1] = -x[j]; }
        break;
    case K_REDFT01:                       /* even about 0, odd about n (mkre01) */
        for (j = 0; j < n; ++j) {
            e[j] = x[j];
            e[2 * n - j] = -x[j];
        }
        e[n] = 0.0;
        for (j = 1; j < 2 * n; ++j) e[n0 - j] = e[j];
        break;
    case K_RODFT01:                       /* odd about 0, even about n (mkro01) */
        for (j = 0; j < n; ++j) {
            e[j + 1] = x[j];
            e[2 * n - j - 1] = x[j];
        }
        for (j = 1; j < 2 * n; ++j) e[n0 - j] = -e[j];
        break;
    case K_REDFT11:                       /* odd points of the 8n grid; odd about 2n, even about 0 (mkre11) */
        for (j = 0; j < n; ++j) {
            e[2 * j + 1] = x[j];
            e[4 * n - 2 * j - 1] = -x[j];
        }
        for (j = 1; j < 4 * n; ++j) e[n0 - j] = e[j];
        break;
    case K_RODFT11:                       /* even about 2n, odd about 0 (mkro11) */
        for (j = 0; j < n; ++j) {
            e[2 * j + 1] = x[j];
            e[4 * n - 2 * j - 1] = x[j];
        }
        for (j = 1; j < 4 * n; ++j) e[n0 - j] = -e[j];
        break;
    }
    r2c_row(n0, e, F);
    switch (kind) {
    case K_R2HC:
        for (k = 0; k <= n / 2; ++k) {
            y[k] = creal(F[k]);
            if (k > 0 && 2 * k < n) y[n - k] = cimag(F[k]);
        }
        break;
    case K_DHT:
        for (k = 0; k <= n / 2; ++k) {
            y[k] = creal(F[k]) - cimag(F[k]);
            if (k > 0 && 2 * k < n) y[n - k] = creal(F[k]) + cimag(F[k]);
        }
        break;
    case K_REDFT00: for (k = 0; k < n; ++k) y[k] = creal(F[k]); break;
    case K_RODFT00: for (k = 0; k < n; ++k) y[k] = -cimag(F[k + 1]); break;
    case K_REDFT10: for (k = 0; k < n; ++k) y[k] = creal(F[k]); break;
    case K_RODFT10: for (k = 0; k < n; ++k) y[k] = -cimag(F[k + 1]); break;
    case K_REDFT01: case K_REDFT11: for (k = 0; k < n; ++k) y[k] = 0.5 * creal(F[2 * k + 1]); break;
    case K_RODFT01: case K_RODFT11: for (k = 0; k < n; ++k) y[k] = -0.5 * cimag(F[2 * k + 1]); break;
    }
    free(e); free(F);
    return 0;
}

/* The same kinds straight from their defining sums, O(n^2), with the exactly
   reduced trig of the reference's verifier (cos00 .. sin11,
   fftw/libbench2/verify-r2r.c:104-146; definitions fftw/doc/reference.texi:1905-2000
   halfcomplex, :2058-2130 DCTs, :2150-2230 DSTs, :2245-2270 DHT).  Cross-check
   for r2r_row. */
int oracle_r2r_direct(int kind, int n_, const double *x, double *y) {
    i64 n = n_, j, k;
    double w[2];
#define COS2PI(m, N) (oracle_cexp((m), (N), w), w[0])
#define SIN2PI(m, N) (oracle_cexp((m), (N), w), w[1])
    for (k = 0; k < n; ++k) {
        long double acc = 0.0L;
        switch (kind) {
        case K_R2HC:
            if (2 * k <= n) { for (j = 0; j < n; ++j) acc += x[j] * COS2PI(j * k, n); }
            else { for (j = 0; j < n; ++j) acc -= x[j] * SIN2PI(j * (n - k), n); }
            break;
        case K_HC2R:
            acc = x[0];
            for (j = 1; 2 * j < n; ++j)
                acc += 2.0L * (x[j] * COS2PI(j * k, n) - x[n - j] * SIN2PI(j * k, n));
            if (n % 2 == 0 && n > 1) acc += x[n / 2] * ((k & 1) ? -1.0 : 1.0);
            break;
        case K_DHT:
            for (j = 0; j < n; ++j) { oracle_cexp(j * k, n, w); acc += x[j] * ((long double)w[0] + w[1]); }
            break;
        case K_REDFT00:
            if (n < 2) return -1;
            acc = x[0] + ((k & 1) ? -x[n - 1] : x[n - 1]);
            for (j = 1; j < n - 1; ++j) acc += 2.0L * x[j] * COS2PI(j * k, 2 * (n - 1));
            break;
        case K_REDFT10:
            for (j = 0; j < n; ++j) acc += 2.0L * x[j] * COS2PI((2 * j + 1) * k, 4 * n);
            break;
        case K_REDFT01:
            acc = x[0];
            for (j = 1; j < n; ++j) acc += 2.0L * x[j] * COS2PI(j * (2 * k + 1), 4 * n);
            break;
        case K_REDFT11:
            for (j = 0; j < n; ++j) acc += 2.0L * x[j] * COS2PI((2 * j + 1) * (2 * k + 1), 8 * n);
            break;
        case K_RODFT00:
            for (j = 0; j < n; ++j) acc += 2.0L * x[j] * SIN2PI((j + 1) * (k + 1), 2 * (n + 1));
            break;
        case K_RODFT10:
            for (j = 0; j < n; ++j) acc += 2.0L * x[j] * SIN2PI((2 * j + 1) * (k + 1), 4 * n);
            break;
        case K_RODFT01:
            acc = (k & 1) ? -x[n - 1] : x[n - 1];
            for (j = 0; j < n - 1; ++j) acc += 2.0L * x[j] * SIN2PI((j + 1) * (2 * k + 1), 4 * n);
            break;
        case K_RODFT11:
            for (j = 0; j < n; ++j) acc += 2.0L * x[j] * SIN2PI((2 * j + 1) * (2 * k + 1), 8 * n);
            break;
        default:
            return -1;
        }
        y[k] = (double)acc;
    }
#undef COS2PI
#undef SIN2PI
    return 0;
}

/* fftw_plan_many_r2r + fftw_execute: kind[d] along dim d, separably
   (reference problem_rdft, fftw/fftw_api.c:9100-9150, API :816-838) */
int oracle_r2r_many(int rank, const int *n, int howmany,
                    const double *in, const int *inembed, int istride, int idist,
                    double *out, const int *onembed, int ostride, int odist, const int *kind) {
    i64 nn[16], ie[16], oe[16], total = 1, f, maxn = 1;
    int d, b, rc = 0;
    double *a, *ti, *to;
    if (rank < 0 || rank > 16 || howmany < 0) return -1;
    cache_gc();
    for (d = 0; d < rank; ++d) {
        if (n[d] <= 0) return -1;
        nn[d] = n[d];
        ie[d] = inembed ? inembed[d] : n[d];
        oe[d] = onembed ? onembed[d] : n[d];
        total *= n[d];
        if (n[d] > maxn) maxn = n[d];
    }
    a = (double *)malloc(sizeof(double) * (size_t)total);
    ti = (double *)malloc(sizeof(double) * (size_t)maxn);
    to = (double *)malloc(sizeof(double) * (size_t)maxn);
    for (b = 0; b < howmany && !rc; ++b) {
        const double *ib = in + (i64)b * idist;
        double *ob = out + (i64)b * odist;
        i64 stride = 1;
        for (f = 0; f < total; ++f) a[f] = ib[embed_off(rank, nn, ie, f, istride)];
        for (d = rank - 1; d >= 0 && !rc; --d) {
            i64 outer = total / (nn[d] * stride), o, s, j;
            for (o = 0; o < outer && !rc; ++o)
                for (s = 0; s < stride && !rc; ++s) {
                    double *base = a + o * nn[d] * stride + s;
                    for (j = 0; j < nn[d]; ++j) ti[j] = base[j * stride];
                    rc = r2r_row(kind[d], nn[d], ti, to);
                    for (j = 0; j < nn[d]; ++j) base[j * stride] = to[j];
                }
            stride *= nn[d];
        }
        for (f = 0; f < total; ++f) ob[embed_off(rank, nn, oe, f, ostride)] = a[f];
    }
    free(a); free(ti); free(to);
    return rc;
}
