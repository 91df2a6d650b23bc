/*
 * hostmath.c -- host numerics of the planner: twiddle values, modular
 * arithmetic for Rader, factorisation into passes and radices.
 *
 * Twiddle accuracy contract: tables are
 * generated on the host in double precision with the argument reduced to
 * [0, pi/4] by octant symmetry before libm is called, which is what the
 * reference does in real_cexp (fftw/fftw_api.c:18850-18892).  The device never
 * evaluates sin/cos.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fa_plan.h"

/* (cos, sin)(2 pi m / n), any integer m.  Octant reduction: fold the angle
   into the first octant, evaluate there, then undo the folds.  All index
   arithmetic is done on 8m versus 8n/… scaled integers so no precision is
   lost before the single division. */
void fa_cexp(i64 m, i64 n, double out[2]) {
    const double two_pi = 6.2831853071795864769252867665590057683943388;
    i64 full = 4 * n;      /* angle unit: 1/(4n) of a turn, so n = quarter turn */
    i64 q = n;             /* quarter turn */
    i64 a;
    int flip_s = 0, rot = 0, swap = 0;
    m %= n;
    if (m < 0) m += n;
    a = 4 * m;             /* in [0, 4n) */
    if (a > full - a) { a = full - a; flip_s = 1; }   /* upper half -> mirror in x axis */
    if (a > q) { a -= q; rot = 1; }                   /* second quadrant -> rotate by 90 deg */
    if (a > q - a) { a = q - a; swap = 1; }           /* above 45 deg -> mirror in diagonal */
    {
        double th = (two_pi * (double)a) / (double)full;
        double c = cos(th), s = sin(th), t;
        if (swap) { t = c; c = s; s = t; }
        if (rot) { t = c; c = -s; s = t; }
        if (flip_s) s = -s;
        out[0] = c;
        out[1] = s;
    }
}

/* overflow-safe (x*y) mod p for 0 <= x,y < p < 2^62 (the reference guards the
   same hazard with MULMOD / safe_mulmod, fftw/fftw_api.c:15750-15766) */
i64 fa_mulmod(i64 x, i64 y, i64 p) {
    return (i64)(((unsigned __int128)x * (unsigned __int128)y) % (unsigned __int128)p);
}

i64 fa_power_mod(i64 b, i64 e, i64 p) {
    i64 r = 1 % p;
    b %= p;
    if (b < 0) b += p;
    while (e > 0) {
        if (e & 1) r = fa_mulmod(r, b, p);
        b = fa_mulmod(b, b, p);
        e >>= 1;
    }
    return r;
}

int fa_prime_factors(i64 n, i64 *primes, int *mult) {
    int k = 0;
    i64 d;
    for (d = 2; d * d <= n; d += (d == 2 ? 1 : 2)) {
        if (n % d == 0) {
            primes[k] = d;
            mult[k] = 0;
            while (n % d == 0) { n /= d; ++mult[k]; }
            ++k;
        }
    }
    if (n > 1) { primes[k] = n; mult[k] = 1; ++k; }
    return k;
}

int fa_is_prime(i64 n) {
    i64 d;
    if (n < 2) return 0;
    for (d = 2; d * d <= n; d += (d == 2 ? 1 : 2))
        if (n % d == 0) return 0;
    return 1;
}

i64 fa_largest_prime_factor(i64 n) {
    i64 pr[64];
    int mu[64];
    int k = fa_prime_factors(n, pr, mu);
    return k ? pr[k - 1] : 1;
}

/* smallest primitive root of the prime p: g is a generator iff
   g^((p-1)/f) != 1 for every prime factor f of p-1 (same test as the
   reference's fftw_find_generator, fftw/fftw_api.c:15812-15828) */
i64 fa_find_generator(i64 p) {
    i64 pr[64];
    int mu[64];
    int k, i;
    i64 g;
    if (p == 2) return 1;
    k = fa_prime_factors(p - 1, pr, mu);
    for (g = 2; g < p; ++g) {
        int ok = 1;
        for (i = 0; i < k; ++i)
            if (fa_power_mod(g, (p - 1) / pr[i], p) == 1) { ok = 0; break; }
        if (ok) return g;
    }
    return 0;
}

/* can the LDS pass kernel transform length n by itself (every prime factor
   has a butterfly or is small enough for the in-LDS O(p^2) stage)? */
int fa_lds_able(i64 n) {
    return n >= 1 && fa_largest_prime_factor(n) <= FA_PRIME_LDS_MAX;
}

/* smallest m >= n whose prime factors are all <= 7 (Bluestein padding; the
   reference pads to a {2,3,5}-smooth size, fftw/fftw_api.c:1738-1742) */
i64 fa_next_smooth(i64 n) {
    for (;; ++n) {
        i64 m = n;
        while (m % 2 == 0) m /= 2;
        while (m % 3 == 0) m /= 3;
        while (m % 5 == 0) m /= 5;
        while (m % 7 == 0) m /= 7;
        if (m == 1) return n;
    }
}

/* ---- radices of one pass --------------------------------------------- */

static int cmp_desc(const void *a, const void *b) {
    int x = *(const int *)a, y = *(const int *)b;
    return (x < y) - (x > y);
}

/* L -> radices; odd primes one stage each, the power of two in balanced
   stages of at most 16.  Largest radix first: the first Stockham stage has
   no twiddles. */
int fa_radices(i64 L, int *rad) {
    int k = 0, e = 0, ns, i;
    i64 m = L;
    i64 p;
    while (m % 2 == 0) { m /= 2; ++e; }
    for (p = 3; m > 1; p += 2) {
        while (m % p == 0) {
            if (k >= FFTW_AMD_MAX_RADICES) return -1;
            rad[k++] = (int)p;
            m /= p;
        }
    }
    ns = (e + 3) / 4;
    for (i = 0; i < ns; ++i) {
        int bits = e / (ns - i);
        if (e % (ns - i)) ++bits;
        if (k >= FFTW_AMD_MAX_RADICES) return -1;
        rad[k++] = 1 << bits;
        e -= bits;
    }
    qsort(rad, (size_t)k, sizeof(int), cmp_desc);
    return k;
}

/* ---- splitting n into passes ------------------------------------------ */

static int cmp_i64(const void *a, const void *b) {
    i64 x = *(const i64 *)a, y = *(const i64 *)b;
    return (x > y) - (x < y);
}

static int divisors_of(i64 n, i64 **out) {
    i64 pr[64];
    int mu[64];
    int k = fa_prime_factors(n, pr, mu);
    int cnt = 1, i, j, c;
    i64 *d;
    for (i = 0; i < k; ++i) cnt *= (mu[i] + 1);
    d = (i64 *)malloc(sizeof(i64) * (size_t)cnt);
    d[0] = 1;
    c = 1;
    for (i = 0; i < k; ++i) {
        int base = c;
        i64 pw = 1;
        for (j = 1; j <= mu[i]; ++j) {
            int t;
            pw *= pr[i];
            for (t = 0; t < base; ++t) d[c++] = d[t] * pw;
        }
    }
    qsort(d, (size_t)cnt, sizeof(i64), cmp_i64);
    *out = d;
    return cnt;
}

static int split_rec(i64 n, int k, i64 lmax, i64 *out) {
    i64 *divs;
    int nd, i, ok = 0;
    double target;
    if (k == 1) {
        if (n <= lmax) { out[0] = n; return 1; }
        return 0;
    }
    target = pow((double)n, 1.0 / (double)k) * (1.0 - 1e-12);
    nd = divisors_of(n, &divs);
    for (i = 0; i < nd && !ok; ++i) {
        i64 d = divs[i];
        if ((double)d < target) continue;
        if (d > lmax) break;
        if (split_rec(n / d, k - 1, lmax, out + 1)) { out[0] = d; ok = 1; }
    }
    free(divs);
    return ok;
}

/* n -> lens[0..k): smallest k <= max_passes with balanced factors, every
   factor <= lmax_multi (k >= 2) or <= lmax_single (k = 1).  0 if impossible. */
int fa_factor_passes(i64 n, int max_passes, i64 lmax_single, i64 lmax_multi, i64 *lens) {
    int k;
    if (n <= lmax_single) { lens[0] = n; return 1; }
    for (k = 2; k <= max_passes; ++k)
        if (split_rec(n, k, lmax_multi, lens)) return k;
    return 0;
}

/* ---- splitting with a preference for lengths that have a register kernel ---- */

typedef struct {
    int k;
    i64 lmax;
    int (*tuned)(i64);
    i64 cur[FA_MAXPASS], best[FA_MAXPASS];
    int best_bad;
    double best_ratio;
    int found;
} pref_state;

static void pref_rec(pref_state *st, i64 n, int depth, const i64 *divs, int nd) {
    int i;
    if (depth == st->k - 1) {
        int j, bad = 0;
        i64 mn, mx;
        if (n > st->lmax) return;
        st->cur[depth] = n;
        mn = mx = st->cur[0];
        for (j = 0; j < st->k; ++j) {
            if (!st->tuned(st->cur[j])) ++bad;
            if (st->cur[j] < mn) mn = st->cur[j];
            if (st->cur[j] > mx) mx = st->cur[j];
        }
        if (mn < 2) return;
        if (!st->found || bad < st->best_bad ||
            (bad == st->best_bad && (double)mx / (double)mn < st->best_ratio)) {
            st->found = 1;
            st->best_bad = bad;
            st->best_ratio = (double)mx / (double)mn;
            memcpy(st->best, st->cur, sizeof(st->cur));
        }
        return;
    }
    for (i = 0; i < nd; ++i) {
        i64 d = divs[i];
        if (d < 2) continue;
        if (d > st->lmax) break;
        if (n % d) continue;
        st->cur[depth] = d;
        pref_rec(st, n / d, depth + 1, divs, nd);
    }
}

/* like fa_factor_passes (same number of passes k), but among all splits into k
   lengths <= lmax_multi pick the one with the fewest lengths that lack a
   register kernel (`tuned` says which have one), then the most balanced. */
int fa_factor_passes_pref(i64 n, int max_passes, i64 lmax_single, i64 lmax_multi, i64 *lens,
                          int (*tuned)(i64)) {
    int k = fa_factor_passes(n, max_passes, lmax_single, lmax_multi, lens);
    pref_state st;
    i64 *divs;
    int nd;
    if (k < 2 || !tuned) return k;
    memset(&st, 0, sizeof(st));
    st.k = k;
    st.lmax = lmax_multi;
    st.tuned = tuned;
    nd = divisors_of(n, &divs);
    pref_rec(&st, n, 0, divs, nd);
    free(divs);
    if (st.found) memcpy(lens, st.best, sizeof(i64) * (size_t)k);
    return k;
}
