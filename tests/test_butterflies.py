"""Host unit test of the device butterflies: fftw3_amd/csrc/butterflies.h is
compiled with g++ (FA_DEV empty, double2 emulated) and every radix is compared
with numpy's DFT.  This checks the straight-line arithmetic only; the kernels
that call it are tested on the GPU."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from util import ROOT

SRC = r'''
struct double2 { double x, y; };
#define FA_DEV inline
#include "butterflies.h"
template <int R> static void run(double *d) {
    cplx x[R];
    for (int i = 0; i < R; ++i) { x[i].x = d[2*i]; x[i].y = d[2*i+1]; }
    Bfly<R>::run(x);
    for (int i = 0; i < R; ++i) { d[2*i] = x[i].x; d[2*i+1] = x[i].y; }
}
extern "C" int bfly(int r, double *d) {
    switch (r) {
    case 2: run<2>(d); break; case 3: run<3>(d); break; case 4: run<4>(d); break;
    case 5: run<5>(d); break; case 7: run<7>(d); break; case 8: run<8>(d); break;
    case 11: run<11>(d); break; case 13: run<13>(d); break; case 16: run<16>(d); break;
    case 15: run<15>(d); break; case 6: run<6>(d); break; case 9: run<9>(d); break;
    case 10: run<10>(d); break; case 12: run<12>(d); break; case 14: run<14>(d); break;
    case 20: run<20>(d); break; case 24: run<24>(d); break; case 25: run<25>(d); break;
    case 18: run<18>(d); break; case 21: run<21>(d); break; case 22: run<22>(d); break;
    case 26: run<26>(d); break; case 27: run<27>(d); break; case 28: run<28>(d); break;
    case 30: run<30>(d); break; case 17: run<17>(d); break; case 19: run<19>(d); break; case 23: run<23>(d); break; case 29: run<29>(d); break; case 31: run<31>(d); break;
    default: return -1;
    }
    return 0;
}
'''


def test_butterflies_match_dft():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "b.cpp")
        so = os.path.join(td, "b.so")
        open(src, "w").write(SRC)
        subprocess.run(["g++", "-O0", "-std=c++17", "-shared", "-fPIC",
                        "-I", os.path.join(ROOT, "fftw3_amd", "csrc"), src, "-o", so], check=True)
        lib = C.CDLL(so)
        lib.bfly.argtypes = [C.c_int, C.c_void_p]
        rng = np.random.default_rng(0)
        for r in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31):
            for _ in range(3):
                x = rng.random(r) - 0.5 + 1j * (rng.random(r) - 0.5)
                d = x.copy()
                assert lib.bfly(r, d.ctypes.data) == 0
                assert np.abs(d - np.fft.fft(x)).max() < 4e-16 * r, r
