"""Test infrastructure: a numpy interpreter of the planner's step list.

It executes, on the CPU, exactly what each kernel launch is *specified* to do
(addressing, flags, twiddle tables, chunk loop) using numpy for the arithmetic.
It exists so that the host logic -- the planner in fftw3_amd/csrc/planner.c --
can be checked in the CPU-only test tier, where no HIP device is present.  It is
never imported by the product (fftw3_amd/) and is not a fallback: the library
itself can only execute on a GPU.
"""
import numpy as np

import fftw3_amd as fa


def _grids(sizes):
    """index grids for the given extents, broadcastable against each other"""
    n = len(sizes)
    out = []
    for k, s in enumerate(sizes):
        shape = [1] * n
        shape[k] = s
        out.append(np.arange(s, dtype=np.int64).reshape(shape))
    return out


def _load(buf, off, im, flags):
    re = buf[off]
    imv = np.zeros_like(re) if (flags & fa.F_REAL_IN) else buf[off + im]
    if flags & fa.F_SWAP_IN:
        re, imv = imv, re
    return re + 1j * imv


def _store(buf, off, im, flags, v):
    if flags & fa.F_CONJ_OUT:
        v = np.conj(v)
    re, imv = v.real, v.imag
    if flags & fa.F_SWAP_OUT:
        re, imv = imv, re
    off = np.broadcast_to(off, v.shape)
    buf[off] = re
    if not (flags & fa.F_REAL_OUT):
        buf[off + im] = imv


class Interp(object):
    def __init__(self, plan):
        self.plan = plan
        self.steps = plan.steps()
        self.tables = {}

    def table(self, tid):
        if tid in self.tables:
            return self.tables[tid]
        t = self.plan.table(tid)
        if isinstance(t, tuple):            # DFT of another table (device-computed in the product)
            src = self.table(t[1])
            t = np.fft.fft(src)
        elif t.dtype == np.float64 and tid in self._perm_ids():
            t = t.astype(np.int64)
        else:
            t = t[0::2] + 1j * t[1::2]
        self.tables[tid] = t
        return t

    def _perm_ids(self):
        ids = set()
        for s in self.steps:
            if s.kind == fa.STEP_COPY and s.table2 >= 0:
                ids.add(s.table2)
        return ids

    def tw2(self, s, m):
        lo = self.table(s.tw_lo)
        hi = self.table(s.tw_hi)
        mask = (1 << s.tw_shift) - 1
        return lo[m & mask] * hi[m >> s.tw_shift]

    def run(self, inbuf, outbuf, scratch_reals):
        """inbuf/outbuf: flat float64 views of the user arrays (may be the same object)."""
        bufs = {0: inbuf, 1: outbuf}
        batch, chunk = self.plan.batch, self.plan.chunk
        if batch == 0:
            return
        for cs in range(0, batch, chunk):
            cn = min(chunk, batch - cs)
            for s in self.steps:
                for b in (s.src_buf, s.dst_buf, s.aux_buf):
                    if b >= 2 and b not in bufs:
                        bufs[b] = np.zeros(scratch_reals, dtype=np.float64)
                self.step(s, bufs, cs, cn)

    def _dims(self, s, cs, cn):
        nd = s.ndims
        dn = [s.dim_n[i] for i in range(nd)]
        dis = [s.dim_is[i] for i in range(nd)]
        dos = [s.dim_os[i] for i in range(nd)]
        dtw = [s.dim_tw[i] for i in range(nd)]
        sbase, dbase = s.src_base, s.dst_base
        bd = s.batch_dim
        if bd >= 0:
            if s.src_buf < 2:
                sbase += cs * dis[bd]
            if s.dst_buf < 2:
                dbase += cs * dos[bd]
            dn[bd] = cn
        if s.kind == fa.STEP_PASS and s.tile_lo_n > 1:     # inner tile component = one more loop
            dn.append(s.tile_lo_n)
            dis.append(s.tile_lo_is)
            dos.append(s.tile_lo_os)
            dtw.append(0)
        return dn, dis, dos, dtw, sbase, dbase

    def step(self, s, bufs, cs, cn):
        dn, dis, dos, dtw, sbase, dbase = self._dims(s, cs, cn)
        src, dst = bufs[s.src_buf], bufs[s.dst_buf]
        if s.kind == fa.STEP_PASS:
            L = s.L
            g = _grids([L] + dn)
            l, idx = g[0], g[1:]
            soff = sbase + l * s.is_l
            doff = dbase + l * s.os_l
            twb = np.zeros_like(l)
            for i, gi in enumerate(idx):
                soff = soff + gi * dis[i]
                doff = doff + gi * dos[i]
                twb = twb + gi * dtw[i]
            rad = [s.radices[i] for i in range(s.nradices)]
            assert int(np.prod(rad)) == L if rad else L == 1
            if s.variant == 7:
                # FFTW_AMD_K_BLUE: Bluestein for a whole row in one kernel -- L = padded length nb, aux_n = n,
                # tw_lo / tw_hi = chirp and kernel tables (pass3b.hpp)
                n = s.aux_n
                assert 2 * n - 1 <= L and s.tw_n == 0
                g2 = _grids([n] + dn)
                so2 = sbase + g2[0] * s.is_l
                do2 = dbase + g2[0] * s.os_l
                for i, gi in enumerate(g2[1:]):
                    so2 = so2 + gi * dis[i]
                    do2 = do2 + gi * dos[i]
                x = _load(src, so2, s.src_im, s.flags & fa.F_SWAP_IN)
                chirp = self.table(s.tw_lo)[:n].reshape([n] + [1] * len(dn))
                kern = self.table(s.tw_hi).reshape([L] + [1] * len(dn))
                b = np.zeros((L,) + x.shape[1:], dtype=np.complex128)
                b[:n] = x * np.conj(chirp)
                c = np.fft.ifft(np.fft.fft(b, axis=0) * kern, axis=0) * L
                _store(dst, do2, s.dst_im, s.flags & fa.F_SWAP_OUT, c[:n] * np.conj(chirp))
                return
            if s.flags & fa.F_REAL_DEC:
                # last trip of a two-trip r2c of n = L1 x L real points (include/fftw3_amd.h): dims[0] = rows
                # k1 = 0 ... L1 / 2 of the scratch image Z[k1][c]; the row's inputs are the separated real-column
                # spectra; outputs k2 < L / 2 in place, the others conjugated at the mirrored index
                assert (s.flags & fa.F_TW_IN) and s.tw_n and dtw[0] == 1 and s.src_im == 1 and s.dst_im == 1
                nrow = dn[0]
                L1 = 2 * (nrow - 1)
                k1 = idx[0]
                km = np.where(k1 == 0, 0, L1 - k1)
                rest_s = np.zeros_like(l)
                rest_d = np.zeros_like(l)
                for i, gi in enumerate(idx):
                    if i:
                        rest_s = rest_s + gi * dis[i]
                        rest_d = rest_d + gi * dos[i]
                c = l >> 1
                o1 = sbase + c * s.is_l + k1 * dis[0] + rest_s
                o2 = sbase + c * s.is_l + km * dis[0] + rest_s
                Z1 = src[o1] + 1j * src[o1 + 1]
                Z2 = src[o2] + 1j * src[o2 + 1]
                x = np.where((l & 1) == 1, -0.5j * (Z1 - np.conj(Z2)), 0.5 * (Z1 + np.conj(Z2)))
                m = l * twb
                assert m.max() < s.tw_n
                y = np.fft.fft(x * np.conj(self.tw2(s, m)), axis=0)
                shape = y.shape
                k2 = np.broadcast_to(l, shape)
                k1b = np.broadcast_to(k1, shape)
                direct = (k2 < L // 2) | ((k1b == 0) & (k2 == L // 2))
                mirror = (k2 >= L // 2) & (k1b != 0) & (2 * k1b != L1)
                y = np.where((k1b == 0) & ((k2 == 0) | (k2 == L // 2)), y.real + 0j, y)
                od = np.broadcast_to(dbase + k2 * s.os_l + k1 * dos[0] + rest_d, shape)
                om = np.broadcast_to(dbase + (L - 1 - k2) * s.os_l + (L1 - k1) * dos[0] + rest_d, shape)
                dst[od[direct]] = y.real[direct]
                dst[od[direct] + 1] = y.imag[direct]
                dst[om[mirror]] = y.real[mirror]
                dst[om[mirror] + 1] = -y.imag[mirror]
                return
            if s.flags & fa.F_C2R_ROWS:
                # fused c2r: L + 1 spectrum entries per row -> 2L reals stored as L (re, im) pairs
                g2 = _grids([L + 1] + dn)
                so2 = sbase + g2[0] * s.is_l
                for i, gi in enumerate(g2[1:]):
                    so2 = so2 + gi * dis[i]
                if s.aux_valid:
                    # fused r2r prologue: the spectrum entries are built from the real input
                    import types
                    sh = types.SimpleNamespace(variant=int(s.aux_valid), is_l=s.is_l, src_im=s.src_im, aux_n=s.aux_n,
                                               tw_lo=s.tw_lo, tw_hi=s.tw_hi, tw_shift=s.tw_shift)
                    Y = self._pro_load(sh, src, so2 - g2[0] * s.is_l, g2[0])
                    Y = np.array(np.broadcast_to(Y, np.broadcast(so2, g2[0]).shape), dtype=np.complex128)
                else:
                    Y = _load(src, so2, s.src_im, 0)
                Y[0] = Y[0].real
                Y[L] = Y[L].real
                xr = np.fft.irfft(Y, n=2 * L, axis=0) * (2 * L)
                if s.aux_buf > 0:
                    # DCT-III / DST-III output shuffle done by the store: y[2j] = v[j], y[2j+1] = +-v[n-1-j]
                    N = 2 * L
                    yy = np.empty_like(xr)
                    yy[0::2] = xr[:L]
                    yy[1::2] = xr[::-1][:L] * (-1.0 if s.aux_buf == fa.R2R_POST_O01 else 1.0)
                    g3 = _grids([N] + dn)
                    do3 = dbase + g3[0] * s.os_l
                    for i, gi in enumerate(g3[1:]):
                        do3 = do3 + gi * dos[i]
                    dst[np.broadcast_to(do3, yy.shape)] = yy
                    return
                _store(dst, doff, s.dst_im, 0, xr[0::2] + 1j * xr[1::2])
                return
            if (s.flags & fa.F_R2C_ROWS) and s.aux_buf > 0:
                # in-row gather of the r2r pre-processing: v[m], m < 2L, from the user's row (stride is_l)
                mode, N = s.aux_buf, 2 * L
                n = {fa.R2R_PRE_E00: N // 2 + 1, fa.R2R_PRE_O00: N // 2 - 1}.get(mode, N)
                row0 = soff - l * s.is_l                              # offset of element 0 of every row
                m = np.arange(N, dtype=np.int64).reshape([N] + [1] * len(dn))
                if mode in (fa.R2R_PRE_E10, fa.R2R_PRE_O10):
                    si = np.where(m < (n + 1) // 2, 2 * m, 2 * n - 1 - 2 * m)
                    v = src[row0[0:1] + si * s.is_l]
                    if mode == fa.R2R_PRE_O10:
                        v = np.where(si & 1, -v, v)
                elif mode == fa.R2R_PRE_E00:
                    v = src[row0[0:1] + np.where(m < n, m, 2 * (n - 1) - m) * s.is_l]
                else:
                    lo = (m >= 1) & (m <= n)
                    hi = m > n + 1
                    v = np.where(lo, src[row0[0:1] + np.where(lo, m - 1, 0) * s.is_l], 0.0) \
                        - np.where(hi, src[row0[0:1] + np.where(hi, 2 * (n + 1) - m - 1, 0) * s.is_l], 0.0)
                x = v[0::2] + 1j * v[1::2]
            else:
                x = _load(src, soff, s.src_im, s.flags)
            if s.tw_n and (s.flags & fa.F_TW_IN):
                m = l * twb
                assert m.max() < s.tw_n
                x = x * np.conj(self.tw2(s, m))
            y = np.fft.fft(x, axis=0)
            if s.flags & fa.F_LO_DFT:
                # the tile's inner component (appended last by _dims) is transformed too: 2-D DFT tile_lo_n x L
                assert s.tile_lo_n > 1 and not s.tw_n
                y = np.fft.fft(y, axis=-1)
            if s.tw_n and not (s.flags & fa.F_TW_IN):
                m = l * twb
                assert m.max() < s.tw_n
                y = y * np.conj(self.tw2(s, m))
            if s.flags & fa.F_R2C_ROWS:
                # fused r2c untangle for n = 2L: L + 1 outputs per row at stride os_l
                zm = np.conj(np.roll(y[::-1], 1, axis=0))            # conj Z[(L - k) % L]
                k = np.arange(L, dtype=np.int64).reshape([L] + [1] * len(dn))
                E = 0.5 * (y + zm)
                O = -0.5j * (y - zm)
                Y = E + O * np.conj(self.tw2(s, k))
                Y[0] = Y[0].real
                nyq = (y[0].real - y[0].imag) + 0j                   # Y[L] = Re Z[0] - Im Z[0]
                dbase_off = doff - l * s.os_l                        # offset of entry 0 of every row
                if s.aux_valid:
                    # fused r2r epilogue on the L + 1 half-spectrum entries (aux_base = twiddle multiplier)
                    import types
                    sh = types.SimpleNamespace(variant=int(s.aux_valid), os_l=s.os_l, dst_im=s.dst_im, flags=0,
                                               aux_n=s.aux_n, tw_lo=s.tw_lo, tw_hi=s.tw_hi, tw_shift=s.tw_shift)
                    Y = E + O * np.conj(self.tw2(s, k * max(1, int(s.aux_base))))
                    Y[0] = Y[0].real
                    self._epi_store(sh, dst, dbase_off[0:1], np.int64(L), nyq[None])
                    self._epi_store(sh, dst, dbase_off, k, Y)
                else:
                    _store(dst, dbase_off + l * s.os_l, s.dst_im, s.flags, Y)
                    _store(dst, (dbase_off + L * s.os_l)[0:1], s.dst_im, s.flags, nyq[None])
            else:
                _store(dst, doff, s.dst_im, s.flags, y)
        elif s.kind in (fa.STEP_COPY, fa.STEP_HERM_EXPAND):
            K = s.aux_n
            g = _grids([K] + dn)
            k, idx = g[0], g[1:]
            soff = np.zeros_like(k) + sbase
            doff = np.zeros_like(k) + dbase
            for i, gi in enumerate(idx):
                soff = soff + gi * dis[i]
                doff = doff + gi * dos[i]
            if s.kind == fa.STEP_HERM_EXPAND:
                half = K // 2
                ks = np.where(k <= half, k, K - k)
                v = _load(src, soff + ks * s.is_l, s.src_im, 0)
                v = np.where(k > half, np.conj(v), v)
                v = np.where((k == 0) | (2 * k == K), v.real + 0j, v)
                _store(dst, doff + k * s.os_l, s.dst_im, s.flags, v)
                return
            ks = k
            if s.flags & fa.F_PERM_SRC:
                ks = self.table(s.table2)[k]
            valid = k < s.aux_valid
            kss = np.where(valid, ks, 0)
            v = _load(src, soff + kss * s.is_l, s.src_im, s.flags)
            v = np.where(valid, v, 0)
            if s.flags & fa.F_MUL_TABLE:
                v = v * self.table(s.table)[k]
            if s.flags & fa.F_MUL_CONJ:
                v = v * np.conj(self.table(s.table)[k])
            kd = k
            if s.flags & fa.F_PERM_DST:
                kd = self.table(s.table2)[k]
            _store(dst, doff + kd * s.os_l, s.dst_im, s.flags, v)
        elif s.kind in (fa.STEP_R2C_POST, fa.STEP_C2R_PRE):
            n = s.aux_n
            h = n // 2
            npair = h // 2 + 1
            g = _grids([npair] + dn)
            k, idx = g[0], g[1:]
            soff = np.zeros_like(k) + sbase
            doff = np.zeros_like(k) + dbase
            for i, gi in enumerate(idx):
                soff = soff + gi * dis[i]
                doff = doff + gi * dos[i]
            km = h - k
            w = self.tw2(s, k * max(1, s.tile))
            if s.kind == fa.STEP_R2C_POST:
                zk = _load(src, soff + k * s.is_l, s.src_im, 0)
                zm = _load(src, soff + np.where(km == h, 0, km) * s.is_l, s.src_im, 0)
                E = 0.5 * (zk + np.conj(zm))
                O = -0.5j * (zk - np.conj(zm))
                P = O * np.conj(w)
                yk = E + P
                ym = np.conj(E - P)
                yk = np.where(k == 0, yk.real + 0j, yk)
                ym = np.where(k == 0, ym.real + 0j, ym)
                # mirrored element first so that k == h-k keeps Y[k]
                self._epi_store(s, dst, doff, km, ym)
                self._epi_store(s, dst, doff, k, yk)
            else:
                yk = self._pro_load(s, src, soff, k)
                ym = self._pro_load(s, src, soff, km)
                yk = np.where(k == 0, yk.real + 0j, yk)
                ym = np.where(k == 0, ym.real + 0j, ym)
                E = yk + np.conj(ym)
                D = yk - np.conj(ym)
                O = D * w
                zk = E + 1j * O
                zm = np.conj(E - 1j * O)
                ok = (km != k) & (km != h)
                # scatter mirrored entries where they exist
                kmm = np.where(ok, km, k)
                _store(dst, doff + kmm * s.os_l, s.dst_im, s.flags, np.where(ok, zm, zk))
                _store(dst, doff + k * s.os_l, s.dst_im, s.flags, zk)
        elif s.kind in (fa.STEP_R2C_POST4, fa.STEP_C2R_PRE4):
            n = s.aux_n
            m = n // 4
            vs = s.aux_valid
            g = _grids([m // 2 + 1] + dn)
            k, idx = g[0], g[1:]
            soff = np.zeros_like(k) + sbase
            doff = np.zeros_like(k) + dbase
            for i, gi in enumerate(idx):
                soff = soff + gi * dis[i]
                doff = doff + gi * dos[i]
            tm = max(1, s.tile)
            w = [None, self.tw2(s, k * tm), self.tw2(s, 2 * k * tm), self.tw2(s, 3 * k * tm)]
            if s.kind == fa.STEP_R2C_POST4:
                km = np.where(k == 0, 0, m - k)
                T = []
                for v in range(2):
                    zk = _load(src, soff + v * vs + k * s.is_l, s.src_im, 0)
                    zm = _load(src, soff + v * vs + km * s.is_l, s.src_im, 0)
                    T.append(0.5 * (zk + np.conj(zm)))
                    T.append(-0.5j * (zk - np.conj(zm)))
                T = [T[0]] + [T[i] * np.conj(w[i]) for i in (1, 2, 3)]
                y0 = T[0] + T[1] + T[2] + T[3]
                y1 = T[0] - 1j * T[1] - T[2] + 1j * T[3]
                y2 = np.conj(T[0] - T[1] + T[2] - T[3])
                y3 = np.conj(T[0] + 1j * T[1] - T[2] - 1j * T[3])
                y0 = np.where(k == 0, y0.real + 0j, y0)
                y2 = np.where(k == 0, y2.real + 0j, y2)
                # duplicates first, the canonical writers last
                self._epi_store(s, dst, doff, m - k, np.where((k == 0), y1, y3))
                self._epi_store(s, dst, doff, 2 * m - k, y2)
                self._epi_store(s, dst, doff, k + m, y1)
                self._epi_store(s, dst, doff, k, y0)
            else:
                Yk = self._pro_load(s, src, soff, k)
                Ykm = self._pro_load(s, src, soff, k + m)
                Y2 = self._pro_load(s, src, soff, 2 * m - k)
                Y1 = self._pro_load(s, src, soff, m - k)
                Yk = np.where(k == 0, Yk.real + 0j, Yk)
                Y2 = np.where(k == 0, Y2.real + 0j, Y2)

                def comb(A, B, Cc, Dc, w1, w2, w3):
                    S0 = A + B + Cc + Dc
                    S1 = A + 1j * B - Cc - 1j * Dc
                    S2 = A - B + Cc - Dc
                    S3 = A - 1j * B - Cc + 1j * Dc
                    return S0 + 1j * (S1 * w1), S2 * w2 + 1j * (S3 * w3)
                z0, z1 = comb(Yk, Ykm, np.conj(Y2), np.conj(Y1), w[1], w[2], w[3])
                z0m, z1m = comb(Y1, Y2, np.conj(Ykm), np.conj(Yk), 1j * np.conj(w[1]), -np.conj(w[2]),
                                -1j * np.conj(w[3]))
                ok = (k != 0) & (2 * k != m)
                km = np.where(ok, m - k, k)
                _store(dst, doff + km * s.os_l, s.dst_im, s.flags, np.where(ok, z0m, z0))
                _store(dst, doff + vs + km * s.os_l, s.dst_im, s.flags, np.where(ok, z1m, z1))
                _store(dst, doff + k * s.os_l, s.dst_im, s.flags, z0)
                _store(dst, doff + vs + k * s.os_l, s.dst_im, s.flags, z1)
        elif s.kind == fa.STEP_RADER_MUL:
            pm1 = s.aux_n
            nvec = int(np.prod(dn)) if dn else 1
            work = bufs[s.src_buf]
            x0b = bufs[s.aux_buf]
            omega = self.table(s.table)
            # destination offsets of Y[0], vectors numbered with dims[0] fastest
            doffs = np.zeros(nvec, dtype=np.int64) + dbase
            rest = np.arange(nvec, dtype=np.int64)
            for i in range(len(dn)):
                doffs += (rest % dn[i]) * dos[i]
                rest //= dn[i]
            for v in range(nvec):
                a = s.src_base + 2 * v * pm1
                A = work[a:a + 2 * pm1:2] + 1j * work[a + 1:a + 2 * pm1:2]
                x0 = x0b[s.aux_base + 2 * v] + 1j * x0b[s.aux_base + 2 * v + 1]
                P = A * omega
                P[0] += x0
                work[a:a + 2 * pm1:2] = P.real
                work[a + 1:a + 2 * pm1:2] = P.imag
                _store(dst, np.array([doffs[v]]), s.dst_im, s.flags, np.array([x0 + A[0]]))
        elif s.kind == fa.STEP_R2R:
            self._r2r(s, src, dst, dn, dis, dos, sbase, dbase)
        else:
            raise AssertionError("unknown step kind %d" % s.kind)

    def _r2r_len(self, s):
        N = s.aux_n
        if s.variant == fa.R2R_POST_E00:
            return N // 2 + 1
        if s.variant == fa.R2R_POST_O00:
            return N // 2 - 1
        return N

    def _epi_store(self, s, dst, doff, idx, Y):
        """store of an untangle step, with the fused r2r epilogue when step.variant names one"""
        mode = s.variant
        if mode == 0:
            _store(dst, doff + idx * s.os_l, s.dst_im, s.flags, Y)
            return
        n = self._r2r_len(s)
        shape = np.broadcast(doff, idx, Y).shape
        doff = np.broadcast_to(doff, shape)
        idx = np.broadcast_to(idx, shape)
        Y = np.broadcast_to(Y, shape)
        mid = (idx > 0) & (2 * idx < n)

        def put(j, v, mask=None):
            j = np.broadcast_to(j, shape)
            v = np.broadcast_to(v, shape)
            if mask is None:
                dst[doff + j * s.os_l] = v
            else:
                dst[(doff + j * s.os_l)[mask]] = v[mask]
        if mode == fa.R2R_POST_R2HC:
            put(idx, Y.real)
            put(n - idx, Y.imag, mid)
        elif mode == fa.R2R_POST_DHT:
            put(idx, np.where(mid, Y.real - Y.imag, Y.real))
            put(n - idx, Y.real + Y.imag, mid)
        elif mode in (fa.R2R_POST_E10, fa.R2R_POST_O10):
            v = (Y.real + 1j * np.where(mid, Y.imag, 0.0)) * np.conj(self.tw2(s, idx))
            if mode == fa.R2R_POST_E10:
                put(idx, 2 * v.real)
                put(n - idx, -2 * v.imag, mid)
            else:
                put(n - 1 - idx, 2 * v.real)
                put(idx - 1, -2 * v.imag, mid)
        elif mode == fa.R2R_POST_E00:
            put(idx, Y.real)
        elif mode == fa.R2R_POST_O00:
            put(idx - 1, -Y.imag, (idx >= 1) & (idx <= n))
        else:
            raise AssertionError("bad fused epilogue %d" % mode)

    def _pro_load(self, s, src, soff, idx):
        """load of a tangle step, with the fused r2r prologue when step.variant names one"""
        mode = s.variant
        if mode == 0:
            return _load(src, soff + idx * s.is_l, s.src_im, 0)
        n = s.aux_n
        shape = np.broadcast(soff, idx).shape
        soff = np.broadcast_to(soff, shape)
        idx = np.broadcast_to(idx, shape)

        def get(j, mask=None):
            if mask is None:
                return src[soff + j * s.is_l]
            return np.where(mask, src[soff + np.where(mask, j, 0) * s.is_l], 0.0)
        if mode == fa.R2R_PRE_HC2R:
            return get(idx) + 1j * get(n - idx, (idx > 0) & (2 * idx < n))
        if mode == fa.R2R_PRE_E01:
            return self.tw2(s, idx) * (get(idx) - 1j * get(n - idx, idx > 0))
        if mode == fa.R2R_PRE_O01:
            return self.tw2(s, idx) * (get(n - 1 - idx) - 1j * get(idx - 1, idx > 0))
        raise AssertionError("bad fused prologue %d" % mode)

    def _r2r(self, s, src, dst, dn, dis, dos, sbase, dbase):
        """FFTW_AMD_STEP_R2R as specified in include/fftw3_amd.h / DESIGN.md section 9"""
        n, K, mode = s.aux_n, s.aux_valid, s.variant
        g = _grids([K] + dn)
        k, idx = g[0], g[1:]
        soff = np.zeros_like(k) + sbase
        doff = np.zeros_like(k) + dbase
        for i, gi in enumerate(idx):
            soff = soff + gi * dis[i]
            doff = doff + gi * dos[i]
        full = np.broadcast(soff, doff).shape
        k = np.broadcast_to(k, full)
        soff = np.broadcast_to(soff, full)
        doff = np.broadcast_to(doff, full)

        def SR(j, mask=None):
            j = np.broadcast_to(j, full)
            if mask is None:
                return src[soff + j * s.is_l]
            return np.where(mask, src[soff + np.where(mask, j, 0) * s.is_l], 0.0)

        def SI(j, mask=None):
            j = np.broadcast_to(j, full)
            if mask is None:
                return src[soff + j * s.is_l + s.src_im]
            return np.where(mask, src[soff + np.where(mask, j, 0) * s.is_l + s.src_im], 0.0)

        def DR(j, v, mask=None):
            j = np.broadcast_to(j, full)
            v = np.broadcast_to(v, full)
            if mask is None:
                dst[doff + j * s.os_l] = v
            else:
                dst[(doff + j * s.os_l)[mask]] = v[mask]

        def SP(j, mask=None):
            """real scratch sequence addressed as pairs: element j at (j >> 1) is_l + (j & 1) src_im"""
            j = np.broadcast_to(j, full)
            jj = j if mask is None else np.where(mask, j, 0)
            v = src[soff + (jj >> 1) * s.is_l + (jj & 1) * s.src_im]
            return v if mask is None else np.where(mask, v, 0.0)

        def DP(j, v, mask=None):
            j = np.broadcast_to(j, full)
            v = np.broadcast_to(v, full)
            addr = doff + (j >> 1) * s.os_l + (j & 1) * s.dst_im
            if mask is None:
                dst[addr] = v
            else:
                dst[addr[mask]] = v[mask]

        def DI(j, v):
            dst[doff + np.broadcast_to(j, full) * s.os_l + s.dst_im] = np.broadcast_to(v, full)

        mid = (k > 0) & (2 * k < n)
        if mode == fa.R2R_PRE_HC2R:
            re, im = SR(k), SR(n - k, mid)
            DR(k, re); DI(k, im)
        elif mode in (fa.R2R_PRE_E10, fa.R2R_PRE_O10):
            has = 2 * k + 1 < n
            b = SR(2 * k + 1, has)
            DP(k, SR(2 * k))
            DP(n - 1 - k, -b if mode == fa.R2R_PRE_O10 else b, has)
        elif mode in (fa.R2R_PRE_E01, fa.R2R_PRE_O01):
            if mode == fa.R2R_PRE_E01:
                x, y = SR(k), SR(n - k, k > 0)
            else:
                x, y = SR(n - 1 - k), SR(k - 1, k > 0)
            w = self.tw2(s, k)
            v = w * (x - 1j * y)
            DR(k, v.real); DI(k, v.imag)
        elif mode == fa.R2R_PRE_E00:
            DP(k, SR(np.where(k < n, k, 2 * (n - 1) - k)))
        elif mode == fa.R2R_PRE_O00:
            N = 2 * (n + 1)
            a = SR(k - 1, (k >= 1) & (k <= n))
            b = SR(N - k - 1, k > n + 1)
            DP(k, a - b)
        elif mode in (fa.R2R_PRE_E11, fa.R2R_PRE_O11):
            xr, xi = SR(2 * k), SR(n - 1 - 2 * k)
            if mode == fa.R2R_PRE_O11:
                xr, xi = xi, xr
            v = (xr + 1j * xi) * np.conj(self.tw2(s, 4 * k))
            DR(k, v.real); DI(k, v.imag)
        elif mode in (fa.R2R_PRE_E11ODD, fa.R2R_PRE_O11ODD):
            lo = k < n
            x = SR(k, lo) if mode == fa.R2R_PRE_E11ODD else SR(n - 1 - k, lo)
            v = x * np.conj(self.tw2(s, np.where(lo, 2 * k, 0)))
            DR(k, v.real); DI(k, v.imag)
        elif mode == fa.R2R_POST_R2HC:
            re, im = SR(k), SI(k, mid)
            DR(k, re)
            DR(n - k, im, mid)
        elif mode == fa.R2R_POST_DHT:
            re, im = SR(k), SI(k, mid)
            DR(k, re - im)
            DR(n - k, re + im, mid)
        elif mode in (fa.R2R_POST_E10, fa.R2R_POST_O10):
            v = (SR(k) + 1j * SI(k, mid)) * np.conj(self.tw2(s, k))
            if mode == fa.R2R_POST_E10:
                DR(k, 2 * v.real)
                DR(n - k, -2 * v.imag, mid)
            else:
                DR(n - 1 - k, 2 * v.real)
                DR(k - 1, -2 * v.imag, mid)
        elif mode in (fa.R2R_POST_E01, fa.R2R_POST_O01):
            has = 2 * k + 1 < n
            b = SP(n - 1 - k, has)
            DR(2 * k, SP(k))
            DR(2 * k + 1, -b if mode == fa.R2R_POST_O01 else b, has)
        elif mode == fa.R2R_POST_E00:
            DR(k, SR(k))
        elif mode == fa.R2R_POST_O00:
            DR(k, -SI(k + 1))
        elif mode in (fa.R2R_POST_E11, fa.R2R_POST_O11):
            v = (SR(k) + 1j * SI(k)) * np.conj(self.tw2(s, 4 * k + 1))
            DR(2 * k, 2 * v.real)
            DR(n - 1 - 2 * k, 2 * v.imag if mode == fa.R2R_POST_O11 else -2 * v.imag)
        elif mode in (fa.R2R_POST_E11ODD, fa.R2R_POST_O11ODD):
            v = (SR(k) + 1j * SI(k)) * np.conj(self.tw2(s, 2 * k + 1))
            y = 2 * v.real
            if mode == fa.R2R_POST_O11ODD:
                y = np.where(k & 1, -y, y)
            DR(k, y)
        else:
            raise AssertionError("unknown r2r mode %d" % mode)


def scratch_reals(plan):
    return max(16, plan.workspace_bytes // 8 + 16)


def run_plan_on_host(plan, inarr, outarr):
    """Interpret `plan` on numpy arrays (float64 views of the user's buffers)."""
    it = Interp(plan)
    a = inarr.reshape(-1).view(np.float64)
    b = a if outarr is inarr else outarr.reshape(-1).view(np.float64)
    it.run(a, b, scratch_reals(plan))
