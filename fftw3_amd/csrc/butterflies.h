/*
 * butterflies.h -- register-resident radix-r DFT butterflies for gfx950 (K7).
 *
 * These are the MI355X counterparts of the reference's no-twiddle codelets
 * n1_2 .. n1_16 (reference fftw/dft_scalar/codelets/n1_*.c; codelet ABI
 * fftw/fftw_api.h:1794-1795): straight-line forward DFTs of r complex doubles
 * held in registers, one butterfly per work-item.  They are hand-written, not
 * generated: radix 2/4/8/16 by explicit Cooley-Tukey splitting, the odd primes
 * 3/5/7/11/13 by the symmetric (x_j +- x_{r-j}) formula that reference
 * fftw/genfft/fft.ml:53-80 documents for prime sizes.
 *
 * Convention: forward transform, X[q] = sum_j x[j] exp(-2 pi i j q / r).
 * Backward transforms are done by the callers with the (re,im) swap identity
 * (reference fftw/fftw_api.c:14555-14564).
 *
 * The header has no HIP dependency beyond FA_DEV/cplx so that the host unit
 * test (tests/test_butterflies.py) can compile it with g++.
 */
#ifndef FA_BUTTERFLIES_H
#define FA_BUTTERFLIES_H

#ifndef FA_DEV
#define FA_DEV __device__ __forceinline__
#endif

#ifndef FA_CPLX_DEFINED
#define FA_CPLX_DEFINED
typedef double2 cplx;
#endif

FA_DEV cplx c_make(double a, double b) { cplx r; r.x = a; r.y = b; return r; }
FA_DEV cplx c_add(cplx a, cplx b) { return c_make(a.x + b.x, a.y + b.y); }
FA_DEV cplx c_sub(cplx a, cplx b) { return c_make(a.x - b.x, a.y - b.y); }
/* a * w */
FA_DEV cplx c_mul(cplx a, cplx w) { return c_make(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
/* a * conj(w): the forward-twiddle product, same form as reference t1_4.c:139-140 */
FA_DEV cplx c_mulc(cplx a, cplx w) { return c_make(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y); }
/* a * (-i) and a * (+i) */
FA_DEV cplx c_mni(cplx a) { return c_make(a.y, -a.x); }
FA_DEV cplx c_mpi(cplx a) { return c_make(-a.y, a.x); }
FA_DEV cplx c_scale(cplx a, double s) { return c_make(a.x * s, a.y * s); }

#define FA_SQRT1_2 0.70710678118654752440084436210484903928483593768847
#define FA_COS_PI_8 0.92387953251128675612818318939678828682241662586364
#define FA_SIN_PI_8 0.38268343236508977172845998403039886676134456248563

template <int R> struct Bfly;

template <> struct Bfly<1> {
    static FA_DEV void run(cplx *) {}
};

template <> struct Bfly<2> {
    static FA_DEV void run(cplx *x) {
        cplx a = x[0];
        x[0] = c_add(a, x[1]);
        x[1] = c_sub(a, x[1]);
    }
};

template <> struct Bfly<4> {
    static FA_DEV void run(cplx *x) {
        cplx a = c_add(x[0], x[2]), b = c_sub(x[0], x[2]);
        cplx c = c_add(x[1], x[3]), d = c_mni(c_sub(x[1], x[3]));
        x[0] = c_add(a, c);
        x[2] = c_sub(a, c);
        x[1] = c_add(b, d);
        x[3] = c_sub(b, d);
    }
};

template <> struct Bfly<8> {
    static FA_DEV void run(cplx *x) {
        cplx e[4] = { x[0], x[2], x[4], x[6] };
        cplx o[4] = { x[1], x[3], x[5], x[7] };
        Bfly<4>::run(e);
        Bfly<4>::run(o);
        /* o[k] *= w8^k, w8 = exp(-i pi/4) */
        cplx t1 = c_make((o[1].x + o[1].y) * FA_SQRT1_2, (o[1].y - o[1].x) * FA_SQRT1_2);
        cplx t2 = c_mni(o[2]);
        cplx t3 = c_make((o[3].y - o[3].x) * FA_SQRT1_2, -(o[3].x + o[3].y) * FA_SQRT1_2);
        x[0] = c_add(e[0], o[0]); x[4] = c_sub(e[0], o[0]);
        x[1] = c_add(e[1], t1);   x[5] = c_sub(e[1], t1);
        x[2] = c_add(e[2], t2);   x[6] = c_sub(e[2], t2);
        x[3] = c_add(e[3], t3);   x[7] = c_sub(e[3], t3);
    }
};

template <> struct Bfly<16> {
    /* 16 = 4 x 4: input j = i + 4 j2, output k = k2 + 4 k1 */
    static FA_DEV void run(cplx *x) {
        cplx z[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            z[i][0] = x[i]; z[i][1] = x[i + 4]; z[i][2] = x[i + 8]; z[i][3] = x[i + 12];
            Bfly<4>::run(z[i]);
        }
        /* twiddle z[i][k2] *= w16^(i k2) */
        const cplx w1 = c_make(FA_COS_PI_8, FA_SIN_PI_8);   /* conj applied by c_mulc */
        const cplx w2 = c_make(FA_SQRT1_2, FA_SQRT1_2);
        const cplx w3 = c_make(FA_SIN_PI_8, FA_COS_PI_8);
        z[1][1] = c_mulc(z[1][1], w1);
        z[1][2] = c_mulc(z[1][2], w2);
        z[1][3] = c_mulc(z[1][3], w3);
        z[2][1] = c_mulc(z[2][1], w2);
        z[2][2] = c_mni(z[2][2]);                            /* w16^4 = -i */
        z[2][3] = c_mulc(z[2][3], c_make(-FA_SQRT1_2, FA_SQRT1_2)); /* w16^6 */
        z[3][1] = c_mulc(z[3][1], w3);
        z[3][2] = c_mulc(z[3][2], c_make(-FA_SQRT1_2, FA_SQRT1_2)); /* w16^6 */
        z[3][3] = c_mulc(z[3][3], c_make(-FA_COS_PI_8, -FA_SIN_PI_8)); /* w16^9 */
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            cplx c[4] = { z[0][k2], z[1][k2], z[2][k2], z[3][k2] };
            Bfly<4>::run(c);
            x[k2] = c[0]; x[k2 + 4] = c[1]; x[k2 + 8] = c[2]; x[k2 + 12] = c[3];
        }
    }
};

/* ---- odd primes: symmetric formula ------------------------------------ */

template <int R> struct OddTrig;
template <> struct OddTrig<3> {
    static FA_DEV double c(int k) { const double t[] = { 1.0, -0.5 }; return t[k]; }
    static FA_DEV double s(int k) { const double t[] = { 0.0, 0.86602540378443864676372317075293618347140262690519 }; return t[k]; }
};
template <> struct OddTrig<5> {
    static FA_DEV double c(int k) {
        const double t[] = { 1.0, 0.30901699437494742410229341718281905886015458990289,
                             -0.80901699437494742410229341718281905886015458990289 };
        return t[k];
    }
    static FA_DEV double s(int k) {
        const double t[] = { 0.0, 0.95105651629515357211643933337938214340569863412575,
                             0.58778525229247312916870595463907276859765243764315 };
        return t[k];
    }
};
template <> struct OddTrig<7> {
    static FA_DEV double c(int k) {
        const double t[] = { 1.0, 0.62348980185873353052500488400423981063227473089640,
                             -0.22252093395631440428890256449679475946635556876452,
                             -0.90096886790241912623610231950744505116591916213185 };
        return t[k];
    }
    static FA_DEV double s(int k) {
        const double t[] = { 0.0, 0.78183148246802980870844452667405775023233451870869,
                             0.97492791218182360701813168299393121723278580062000,
                             0.43388373911755812047576833284835875460999072778746 };
        return t[k];
    }
};
template <> struct OddTrig<11> {
    static FA_DEV double c(int k) {
        const double t[] = { 1.0, 0.84125353283118116886181164892548537747900416427459,
                             0.41541501300188642553467186187277594354271271181612,
                             -0.14231483827328514044379266862004953478227925421141,
                             -0.65486073394528506405692507247051100153382876839305,
                             -0.95949297361449738989036805707015078470561785768624 };
        return t[k];
    }
    static FA_DEV double s(int k) {
        const double t[] = { 0.0, 0.54064081745559758210763595432149119841854573665417,
                             0.90963199535451837141171538308025438115339826461594,
                             0.98982144188093273237609203778476468501456787301956,
                             0.75574957435425828377403584396999779705476123571770,
                             0.28173255684142969771141791534956274523286476786253 };
        return t[k];
    }
};
template <> struct OddTrig<13> {
    static FA_DEV double c(int k) {
        const double t[] = { 1.0, 0.88545602565320989589618721428105947001195879239484,
                             0.56806474673115580251180755912752337437587109759741,
                             0.12053668025532305334906768745253665704210710478091,
                             -0.35460488704253562596963789260002222810827551844950,
                             -0.74851074817110109863463059970135073301528603526167,
                             -0.97094181742605202715698227629378922724986131856198 };
        return t[k];
    }
    static FA_DEV double s(int k) {
        const double t[] = { 0.0, 0.46472317204376854565630555298924221625422258894859,
                             0.82298386589365639457961001233239279374986520735366,
                             0.99270887409805399280075894973008888491603537321479,
                             0.93501624268541482343905466707503480354517331332141,
                             0.66312265824079520237678549965245427733855323887033,
                             0.23931566428755776714875372626641424363111980449113 };
        return t[k];
    }
};

template <int R> struct BflyOdd {
    static FA_DEV void run(cplx *x) {
        constexpr int H = (R - 1) / 2;
        cplx a[H + 1], b[H + 1];
        cplx x0 = x[0];
        cplx sum = x0;
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            a[j] = c_add(x[j], x[R - j]);
            b[j] = c_sub(x[j], x[R - j]);
            sum = c_add(sum, a[j]);
        }
        x[0] = sum;
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            double ur = x0.x, ui = x0.y, vr = 0.0, vi = 0.0;
#pragma unroll
            for (int j = 1; j <= H; ++j) {
                int m = (j * q) % R;           /* compile-time after unrolling */
                double cs = OddTrig<R>::c(m <= H ? m : R - m);
                double sn = OddTrig<R>::s(m <= H ? m : R - m);
                if (m > H) sn = -sn;
                ur += cs * a[j].x; ui += cs * a[j].y;
                vr += sn * b[j].x; vi += sn * b[j].y;
            }
            /* X[q] = u - i v ; X[R-q] = u + i v */
            x[q] = c_make(ur + vi, ui - vr);
            x[R - q] = c_make(ur - vi, ui + vr);
        }
    }
};

template <> struct Bfly<3> : BflyOdd<3> {};
template <> struct Bfly<5> : BflyOdd<5> {};
template <> struct Bfly<7> : BflyOdd<7> {};
template <> struct Bfly<11> : BflyOdd<11> {};
template <> struct Bfly<13> : BflyOdd<13> {};


/* radix 15 = 3 x 5 (input j = i + 3 j2, output k = k2 + 5 k1), constants w15^(i k2) */
template <> struct Bfly<15> {
    static FA_DEV void run(cplx *x) {
        cplx z[3][5];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 5; ++j) z[i][j] = x[i + 3 * j];
            Bfly<5>::run(z[i]);
        }
        const cplx w1 = c_make(0.91354545764260089550, 0.40673664307580020775);
        const cplx w2 = c_make(0.66913060635885821383, 0.74314482547739423501);
        const cplx w3 = c_make(0.30901699437494742410, 0.95105651629515357212);
        const cplx w4 = c_make(-0.10452846326765347140, 0.99452189536827333692);
        const cplx w6 = c_make(-0.80901699437494742410, 0.58778525229247312917);
        const cplx w8 = c_make(-0.97814760073380563793, -0.20791169081775933710);
        z[1][1] = c_mulc(z[1][1], w1); z[1][2] = c_mulc(z[1][2], w2);
        z[1][3] = c_mulc(z[1][3], w3); z[1][4] = c_mulc(z[1][4], w4);
        z[2][1] = c_mulc(z[2][1], w2); z[2][2] = c_mulc(z[2][2], w4);
        z[2][3] = c_mulc(z[2][3], w6); z[2][4] = c_mulc(z[2][4], w8);
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) {
            cplx c[3] = { z[0][k2], z[1][k2], z[2][k2] };
            Bfly<3>::run(c);
            x[k2] = c[0]; x[k2 + 5] = c[1]; x[k2 + 10] = c[2];
        }
    }
};

/* ---- composite radices ---------------------------------------------------- */
/* Coprime factors: the prime-factor (Good-Thomas) map needs no twiddles, only
   compile-time index permutations:  j = (j1 B + j2 A) mod N on the input,
   k = (k1 e1 + k2 e2) mod N on the output with e1 = B (B^-1 mod A), e2 = A (A^-1 mod B)
   (the reference's generator makes the same choice for coprime sizes,
   fftw/genfft/fft.ml:283-300 "prime factor"). */
constexpr int fa_modinv(int a, int m) {
    int r = 1;
    for (int i = 1; i < m; ++i) if ((a * i) % m == 1) r = i;
    return r;
}
template <int A, int B> struct BflyPFA {
    static constexpr int N = A * B;
    static constexpr int E1 = B * fa_modinv(B % A, A);
    static constexpr int E2 = A * fa_modinv(A % B, B);
    static FA_DEV void run(cplx *x) {
        cplx z[A][B];
#pragma unroll
        for (int j1 = 0; j1 < A; ++j1) {
#pragma unroll
            for (int j2 = 0; j2 < B; ++j2) z[j1][j2] = x[(j1 * B + j2 * A) % N];
            Bfly<B>::run(z[j1]);
        }
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) {
            cplx c[A];
#pragma unroll
            for (int j1 = 0; j1 < A; ++j1) c[j1] = z[j1][k2];
            Bfly<A>::run(c);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) x[(k1 * E1 + k2 * E2) % N] = c[k1];
        }
    }
};
template <> struct Bfly<6> : BflyPFA<2, 3> {};
template <> struct Bfly<10> : BflyPFA<2, 5> {};
template <> struct Bfly<12> : BflyPFA<4, 3> {};
template <> struct Bfly<14> : BflyPFA<2, 7> {};
template <> struct Bfly<20> : BflyPFA<4, 5> {};
template <> struct Bfly<24> : BflyPFA<8, 3> {};

/* Squares of odd primes: Cooley-Tukey A x A (input j = i + A j2, output k = k2 + A k1)
   with the constants w_N^(i k2) */
template <int N> struct CtTw;
template <> struct CtTw<9> {
    static FA_DEV cplx w(int k) {
        const cplx t[5] = {
            c_make(1.0, 0.0),
            /* w9^1 */ c_make(0.7660444431189780352023926505554166739358, 0.6427876096865393263226434099072634329076),
            /* w9^2 */ c_make(0.1736481776669303488517166267693147960004, 0.9848077530122080593667430245895230136706),
            c_make(-0.5, 0.8660254037844386467637231707529361834714),
            /* w9^4 */ c_make(-0.9396926207859083840541092773247314699362, 0.3420201433256687330440996146822595807631) };
        return t[k];
    }
};
template <> struct CtTw<25> {
    static FA_DEV cplx w(int k) {
        switch (k) {
        case 1: return c_make(0.9685831611286311194901683754647358138360, 0.2486898871648547882422837460064479684176);
        case 2: return c_make(0.8763066800438635873081159039220625833991, 0.4817536741017152749871915028721296535285);
        case 3: return c_make(0.7289686274214115231467303190552591113726, 0.6845471059286886737322833576212092698895);
        case 4: return c_make(0.5358267949789966182713087678676399780636, 0.8443279255020150785485580639666815053817);
        case 6: return c_make(0.0627905195293133760761782245656311331225, 0.9980267284282715619523368068634505533369);
        case 8: return c_make(-0.4257792915650726488625024457442517039800, 0.9048270524660195277136686479326975939704);
        case 9: return c_make(-0.6374239897486897101767128116760161954349, 0.7705132427757892308030096363961778472717);
        case 12: return c_make(-0.9921147013144778310497930427857785214530, 0.1253332335643042453731187598165087939429);
        case 16: return c_make(-0.6374239897486897101767128116760161954349, -0.7705132427757892308030096363961778472717);
        default: return c_make(1.0, 0.0);
        }
    }
};
template <int A> struct BflySquare {
    static constexpr int N = A * A;
    static FA_DEV void run(cplx *x) {
        cplx z[A][A];
#pragma unroll
        for (int i = 0; i < A; ++i) {
#pragma unroll
            for (int j = 0; j < A; ++j) z[i][j] = x[i + A * j];
            Bfly<A>::run(z[i]);
        }
#pragma unroll
        for (int i = 1; i < A; ++i)
#pragma unroll
            for (int k2 = 1; k2 < A; ++k2) z[i][k2] = c_mulc(z[i][k2], CtTw<N>::w(i * k2));
#pragma unroll
        for (int k2 = 0; k2 < A; ++k2) {
            cplx c[A];
#pragma unroll
            for (int i = 0; i < A; ++i) c[i] = z[i][k2];
            Bfly<A>::run(c);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) x[k2 + A * k1] = c[k1];
        }
    }
};
template <> struct Bfly<9> : BflySquare<3> {};
template <> struct Bfly<25> : BflySquare<5> {};


/* radix 27 = 3 x 9 Cooley-Tukey (input j = i + 3 j2, output k = k2 + 9 k1), constants w27^(i k2) */
template <> struct CtTw<27> {
    static FA_DEV cplx w(int k) {
        switch (k) {
        case 1: return c_make(0.9730448705798238388328851727846959200349, 0.2306158707424401784501983492929391024576);
        case 2: return c_make(0.8936326403234122481925741868666551173761, 0.4487991802004621727850403347331436164243);
        case 3: return c_make(0.7660444431189780352023926505554166739358, 0.6427876096865393263226434099072634329076);
        case 4: return c_make(0.5971585917027861648518521605839597728407, 0.8021231927550437850832948919339251336279);
        case 5: return c_make(0.3960797660391568236960433916097445675085, 0.9182161068802740147589614153146366024814);
        case 6: return c_make(0.1736481776669303488517166267693147960004, 0.9848077530122080593667430245895230136706);
        case 7: return c_make(-0.05814482891047582853874801684707152363411, 0.9983081582712682080478207087832775329371);
        case 8: return c_make(-0.2868032327110902531032801731671579370202, 0.957989512315488874437374766956754624258);
        case 10: return c_make(-0.6862416378687335857296049996175379830146, 0.7273736415730486959871764176638155218004);
        case 12: return c_make(-0.9396926207859083840541092773247314699362, 0.3420201433256687330440996146822595807631);
        case 14: return c_make(-0.9932383577419429885478955521937043403491, -0.1160929141252302296756665233807114688535);
        case 16: return c_make(-0.835487811412936419653826170019583593742, -0.5495089780708060352627803740501339165127);
        default: return c_make(1.0, 0.0);
        }
    }
};
template <int A, int B> struct BflyCT {
    static constexpr int N = A * B;
    static FA_DEV void run(cplx *x) {
        cplx z[A][B];
#pragma unroll
        for (int i = 0; i < A; ++i) {
#pragma unroll
            for (int j = 0; j < B; ++j) z[i][j] = x[i + A * j];
            Bfly<B>::run(z[i]);
        }
#pragma unroll
        for (int i = 1; i < A; ++i)
#pragma unroll
            for (int k2 = 1; k2 < B; ++k2) z[i][k2] = c_mulc(z[i][k2], CtTw<N>::w(i * k2));
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) {
            cplx c[A];
#pragma unroll
            for (int i = 0; i < A; ++i) c[i] = z[i][k2];
            Bfly<A>::run(c);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) x[k2 + B * k1] = c[k1];
        }
    }
};
template <> struct Bfly<27> : BflyCT<3, 9> {};
/* more coprime products (prime-factor maps, no twiddles): the reference has n1_/t1_ codelets up to 64
   for powers of two only; these fill the register-kernel menu for 7-smooth lengths */
template <> struct Bfly<18> : BflyPFA<2, 9> {};
template <> struct Bfly<21> : BflyPFA<3, 7> {};
template <> struct Bfly<22> : BflyPFA<2, 11> {};
template <> struct Bfly<26> : BflyPFA<2, 13> {};
template <> struct Bfly<28> : BflyPFA<4, 7> {};
template <> struct Bfly<30> : BflyPFA<2, 15> {};

/* primes 17, 19, 23: the same symmetric O(p^2) formula (the reference has no codelets for them and runs its
   generic O(n^2) solver, fftw/fftw_api.c:3390-3448); as radices of the two-stage kernels they take lengths such as
   136 = 17 x 8 or 17 408 = 128 x 136 off the LDS kernel */
template <> struct OddTrig<17> {
    static FA_DEV double c(int k) {
        const double t[] = { 1.0,
                             0.9324722294043558045731158918215633862626,
                             0.7390089172206591159245343098726481057599,
                             0.4457383557765382673964575493794868554277,
                             0.09226835946330199523965110715450648036302,
                             -0.2736629900720828635390779354368134316249,
                             -0.602634636379256389178588154986840621619,
                             -0.8502171357296141521341439229493520584707,
                             -0.9829730996839017782819488448551987160987 };
        return t[k];
    }
    static FA_DEV double s(int k) {
        const double t[] = { 0.0,
                             0.3612416661871529487447145961837001637245,
                             0.6736956436465572117126919124256946158624,
                             0.8951632913550623220670164997537854569906,
                             0.9957341762950345218711911789054817839027,
                             0.9618256431728190704087962907315185500315,
                             0.7980172272802395033328051127962613693613,
                             0.526432162877355800244607799140699566171,
                             0.1837495178165703315744088396207275824891 };
        return t[k];
    }
};
template <> struct Bfly<17> : BflyOdd<17> {};
template <> struct OddTrig<19> {
    static FA_DEV double c(int k) {
        const double t[] = { 1.0,
                             0.9458172417006346790196657142849415278238,
                             0.7891405093963935992189811493990907424327,
                             0.5469481581224268747117627466961884997789,
                             0.2454854871407991489222909177963705562718,
                             -0.08257934547233232460034393423744022769858,
                             -0.4016954246529694575168416597426171522567,
                             -0.6772815716257410747621509844956257184155,
                             -0.8794737512064890713908547548818411172079,
                             -0.9863613034027223736025091948190671107285 };
        return t[k];
    }
    static FA_DEV double s(int k) {
        const double t[] = { 0.0,
                             0.3246994692046834874075727165465870379355,
                             0.6142127126896678174443358335144494567519,
                             0.8371664782625285748060612009369102474987,
                             0.9694002659393304167361073217961682259573,
                             0.9965844930066698498193520007504877187805,
                             0.9157733266550574399193492356940089700767,
                             0.7357239106731316247742076119610924993214,
                             0.4759473930370735444313529194551153377644,
                             0.1645945902807338941436520590879384195122 };
        return t[k];
    }
};
template <> struct Bfly<19> : BflyOdd<19> {};
template <> struct OddTrig<23> {
    static FA_DEV double c(int k) {
        const double t[] = { 1.0,
                             0.962917287347799295015223597373238799355,
                             0.8544194045464885525482156195502508000479,
                             0.6825531432186540828745375453725405780988,
                             0.4600650377311521260415757598109517955579,
                             0.2034560130526337898780287220615784267778,
                             -0.06824241336467097592118847902245902393309,
                             -0.3348796121709861519581150708478901575074,
                             -0.5766803221148671412510482752668528239789,
                             -0.7757112907044198070411010109695368955877,
                             -0.9172113015054530178438054479656154936903,
                             -0.99068594603633075234232296009620600514 };
        return t[k];
    }
    static FA_DEV double s(int k) {
        const double t[] = { 0.0,
                             0.2697967711570242712453285226025705364753,
                             0.5195839500354335781330010113237876331493,
                             0.7308359642781241016508331160835884644009,
                             0.8878852184023752349842692774195844835989,
                             0.9790840876823228756328148847602371349847,
                             0.9976687691905391984535782806992783166368,
                             0.9422609221188204956176842253179721336254,
                             0.8169698930104420169734140372449881772468,
                             0.6310879443260527893674001301433105742008,
                             0.39840108984624145799788039996967896565,
                             0.1361666490962465907607258333878729914504 };
        return t[k];
    }
};
template <> struct Bfly<23> : BflyOdd<23> {};

/* 29 and 31: only the one-stage rows kernel (pass1r.hpp) uses them */
template <> struct OddTrig<29> {
    static FA_DEV double c(int k) {
        const double t[] = { 1.0,
                             0.976620555710086683208227962877863351799,
                             0.9075754196709570536201612900285178073502,
                             0.7960930657056437459980762465098682421823,
                             0.6473862847818276391816601341861462687573,
                             0.4684084406997901392162396741494573562814,
                             0.2675283385292208211946262052833413401837,
                             0.05413890858541752614990832597459869261258,
                             -0.1617819965527647265442600643364213138442,
                             -0.370138155339914356863980667615164457098,
                             -0.5611870653623823692699409283736092029758,
                             -0.7259954919231308581383348989285119089043,
                             -0.8568571761675892445230765519053744460274,
                             -0.9476531711828024442740040119711601634623,
                             -0.9941379571543596089553027158795515668546 };
        return t[k];
    }
    static FA_DEV double s(int k) {
        const double t[] = { 0.0,
                             0.2149704402110240671819534770820757537978,
                             0.4198891015602645769737108950291563357024,
                             0.6051742151937651659242801329801084792646,
                             0.7621620551276364632557304138001066169968,
                             0.8835120444460229228273168942218641218896,
                             0.9635499925192229600433361810024919509632,
                             0.9985334138511238645717905110783489569243,
                             0.9868265225415261517686243504388935079839,
                             0.9289767198167914417896296010855542620842,
                             0.8276889981568905561357816231375032629305,
                             0.6876994588534232930838768523753670644636,
                             0.5155538571770217397098664966397134305305,
                             0.3193015301359799731972335422795273269787,
                             0.1081190184239417630308083269836870058627 };
        return t[k];
    }
};
template <> struct Bfly<29> : BflyOdd<29> {};
template <> struct OddTrig<31> {
    static FA_DEV double c(int k) {
        const double t[] = { 1.0,
                             0.9795299412524944939380064428117707242914,
                             0.9189578116202306291271881732781545512765,
                             0.8207634412072763263635445613553707767235,
                             0.68896691907568656780086680381814168713,
                             0.5289640103269624573654923939122347256678,
                             0.347305252844820285541854355481012246462,
                             0.1514277775045766636574676467272196523058,
                             -0.05064916883871271227875185748519952674658,
                             -0.2506525322587205393148020352659594949329,
                             -0.4403941515576343095161715337137760630174,
                             -0.6121059825476628441467056202598600662487,
                             -0.7587581226927909019132546363634371874187,
                             -0.8743466161445821188274846642006517855751,
                             -0.9541392564000488514758967202113007469136,
                             -0.994869323391895146321353309883719493004 };
        return t[k];
    }
    static FA_DEV double s(int k) {
        const double t[] = { 0.0,
                             0.2012985200886600791415289683390134818534,
                             0.3943558551133185801016261030214455736356,
                             0.5712682150947922791574245436284554823535,
                             0.7247927872291199588654846624405482525919,
                             0.8486442574947509504641043389938084539826,
                             0.9377521321470804584291761743123298881309,
                             0.988468324328111399162190689403153774921,
                             0.9987165071710528071463114367595140457475,
                             0.9680771188662043051530076728012907428348,
                             0.8978045395707416571368028976620412024435,
                             0.7907757369376985820782204594612615906186,
                             0.651372482722222207453999614691016466092,
                             0.4853019625310810252145722292597299794313,
                             0.2993631229733579540081126169766754622405,
                             0.1011683219874321777860407155854228233862 };
        return t[k];
    }
};
template <> struct Bfly<31> : BflyOdd<31> {};

#endif /* FA_BUTTERFLIES_H */
