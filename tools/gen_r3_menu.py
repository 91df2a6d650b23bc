#!/usr/bin/env python3
"""Choose the (R1, R2, R3) factorisation of every three-stage rows-kernel length.

Every 7-smooth L in (512, 4096] that is not a power of two and has no two-stage kernel
of its own is factored into three radices <= 32 that have a register butterfly; the most
balanced factorisations are compiled for gfx950 in every order and the one with the
fewest spilled VGPRs (then the most rows per tile) wins.  Writes
fftw3_amd/csrc/r3_menu.inc (committed; the library build never runs this search).

usage: python tools/gen_r3_menu.py            (needs hipcc; a few minutes on 8 cores)
       python tools/gen_r3_menu.py --extend   keep the menu, add 13-smooth lengths and the lengths in (4096, 8192]
"""
import itertools
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fftw3_amd", "csrc")
RADICES = [4, 5, 6, 7, 8, 9, 10, 12, 14, 15, 16, 20, 24, 25, 32]
MAX_SPILL = 4


def smooth7(n):
    for p in (2, 3, 5, 7):
        while n % p == 0:
            n //= p
    return n == 1


def q(nb):
    return (nb + 255) // 256


def tile(r1, r2, r3):
    L = r1 * r2 * r3
    t = max(1, 8192 // L)
    while t > 1 and (q(t * r2 * r3) * r1 > 32 or q(t * r1 * r3) * r2 > 32 or q(t * r1 * r2) * r3 > 32):
        t -= 1
    return t


def fits(r1, r2, r3):
    t = tile(r1, r2, r3)
    return q(t * r2 * r3) * r1 <= 32 and q(t * r1 * r3) * r2 <= 32 and q(t * r1 * r2) * r3 <= 32


def rr_menu():
    out = set()
    with open(os.path.join(CSRC, "rr_menu.inc")) as f:
        for m in re.finditer(r"X\((\d+),", f.read()):
            out.add(int(m.group(1)))
    return out


T_VARIANTS = [("true", 0), ("true", 1), ("false", 0), ("false", 2)]


def compile_triples(idx, triples, strided=False):
    src = os.path.join("/tmp", "r3_menu_%s%d.hip" % ("t" if strided else "", idx))
    with open(src, "w") as f:
        f.write('#include "common.hpp"\n#include "pass1024.hpp"\n#include "passrr.hpp"\n#include "pass3s.hpp"\n'
                '#include "pass3g.hpp"\n')
        for t in triples:
            if strided:
                for it, tw in T_VARIANTS:
                    f.write("template __global__ void pass3t_kernel<%d, %d, %d, %s, %d>(const P1024Args);\n"
                            % (t + (it, tw)))
            else:
                f.write("template __global__ void pass3g_kernel<%d, %d, %d>(const P3SArgs);\n" % t)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
           "-I", CSRC, "-c", src, "-o", src + ".o", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    res, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"Function Name: _Z13pass3[gt]_kernelILi(\d+)ELi(\d+)ELi(\d+)E", line)
        if m:
            cur = tuple(int(v) for v in m.groups())
            continue
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and cur:
            res[cur] = max(res.get(cur, 0), int(m.group(1)))
    return res


def fits_t(r1, r2, r3):
    L = r1 * r2 * r3
    t = 8192 // L
    # t >= 8: 128-byte segments.  Lengths in (1024, 2048] get 4 ... 7 sequences per tile (64 ... 112-byte
    # segments, ~3 TB/s): slower per trip, but the alternative for a strided axis of such a length (the 1080 of
    # a 1080 x 1920 image) is the runtime-radix LDS kernel at 1-2 TB/s
    return t >= 4 and q(t * r2 * r3) * r1 <= 40 and q(t * r1 * r3) * r2 <= 40 and q(t * r1 * r2) * r3 <= 40


RADICES_WIDE = [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 21, 22, 24, 25, 26, 27, 28, 30, 32]


def smooth13(n):
    for p in (2, 3, 5, 7, 11, 13):
        while n % p == 0:
            n //= p
    return n == 1


def extend():
    """--extend: keep every entry of r3_menu.inc and add (a) the 13-smooth lengths in (512, 4096] that have neither
    a two-stage nor a three-stage kernel yet and (b) the lengths in (4096, 8192] (one row per workgroup), over the
    wider radix set; per length the two most balanced factorisations, outer radices as equal as possible.  The
    strided menu (r3t_menu.inc) is not touched."""
    have2 = rr_menu()
    path = os.path.join(CSRC, "r3_menu.inc")
    with open(path) as f:
        old = [tuple(int(v) for v in m.groups()) for m in re.finditer(r"X\((\d+), (\d+), (\d+), (\d+)\)", f.read())]
    have3 = set(e[0] for e in old)
    cands = []
    for L in range(513, 8193):
        if not smooth13(L) or L & (L - 1) == 0 or L in have2 or L in have3:
            continue
        facts = set()
        for a in RADICES_WIDE:
            for b in RADICES_WIDE:
                if L % (a * b) == 0 and (L // (a * b)) in RADICES_WIDE and a <= b <= L // (a * b):
                    facts.add((a, b, L // (a * b)))
        for f in sorted(facts, key=lambda f: f[2] / f[0])[:2]:
            perms = sorted((p for p in set(itertools.permutations(f)) if fits(*p)),
                           key=lambda p: (abs(p[0] - p[2]), p[0] > p[2]))
            cands.extend(perms[:2])
    cands = sorted(set(cands))
    print("%d candidate kernels" % len(cands), file=sys.stderr)
    nproc = 8
    spills = {}
    # small compilation units: a translation unit of 100 big kernels takes minutes
    units = [cands[i:i + 12] for i in range(0, len(cands), 12)]
    with ThreadPoolExecutor(nproc) as ex:
        for r in ex.map(lambda t: compile_triples(*t), enumerate(units)):
            spills.update(r)
    best = {}
    for (a, b, c), s in sorted(spills.items()):
        L = a * b * c
        if s > MAX_SPILL:
            continue
        key = (s, -tile(a, b, c), abs(a - c))
        if L not in best or key < best[L][3]:
            best[L] = (a, b, c, key)
    rows = dict((e[0], e[1:]) for e in old)
    for L in best:
        rows[L] = best[L][:3]
    with open(path, "w") as f:
        f.write("/* generated by tools/gen_r3_menu.py -- X(L, R1, R2, R3): three-stage rows kernel of length L */\n")
        for L in sorted(rows):
            f.write("X(%d, %d, %d, %d)\n" % ((L,) + tuple(rows[L])))
    print("%d new lengths, %d in all" % (len(best), len(rows)), file=sys.stderr)


def real_subset():
    """--real: r3r_menu.inc, the lengths whose rows kernel is also built in its fused real forms (r2c untangle /
    c2r tangle, pass3g_kernel MODE 1 / 2): the 5-smooth entries of r3_menu.inc -- half lengths of the real sizes
    people use (n = 1280 ... 15360: 1920, 2000, 3000, 3600, 4000, 6000, 7200, 10000, 12000 ...)."""
    with open(os.path.join(CSRC, "r3_menu.inc")) as f:
        rows = [tuple(int(v) for v in m.groups()) for m in re.finditer(r"X\((\d+), (\d+), (\d+), (\d+)\)", f.read())]

    def smooth5(n):
        for p in (2, 3, 5):
            while n % p == 0:
                n //= p
        return n == 1
    rows = [r for r in rows if smooth5(r[0])]
    with open(os.path.join(CSRC, "r3r_menu.inc"), "w") as f:
        f.write("/* generated by tools/gen_r3_menu.py --real -- X(L, R1, R2, R3): rows kernels of r3_menu.inc that are\n"
                "   also built with the fused r2c untangle / c2r tangle (half length L of real rows of 2L) */\n")
        for r in rows:
            f.write("X(%d, %d, %d, %d)\n" % r)
    print("%d real-rows lengths" % len(rows), file=sys.stderr)


def blue_ladder():
    """--blue: blue_menu.inc, the padded lengths the one-kernel Bluestein (pass3b.hpp) is built for: a ladder through
    r3_menu.inc in steps of about 4.5 % (a length n takes the first entry >= 2n - 1)."""
    with open(os.path.join(CSRC, "r3_menu.inc")) as f:
        rows = sorted(tuple(int(v) for v in m.groups()) for m in re.finditer(r"X\((\d+), (\d+), (\d+), (\d+)\)", f.read()))
    sel, last = [], 0
    for r in rows:
        if r[0] >= last * 1.045:
            sel.append(r)
            last = r[0]
    if rows[-1] not in sel:
        sel.append(rows[-1])
    sel.append((8192, 32, 16, 16))          # the power of two on top: lengths up to 4096
    with open(os.path.join(CSRC, "blue_menu.inc"), "w") as f:
        f.write("/* generated by tools/gen_r3_menu.py --blue -- X(NB, R1, R2, R3): padded lengths of the one-kernel Bluestein\n"
                "   (pass3b.hpp): a ladder through r3_menu.inc in steps of about 4.5 % */\n")
        for r in sel:
            f.write("X(%d, %d, %d, %d)\n" % r)
    print("%d padded lengths" % len(sel), file=sys.stderr)


def main():
    if "--extend" in sys.argv:
        return extend()
    if "--blue" in sys.argv:
        return blue_ladder()
    if "--real" in sys.argv:
        return real_subset()
    have2 = rr_menu()
    cands = []
    for L in range(513, 4097):
        if not smooth7(L) or L & (L - 1) == 0 or L in have2:
            continue
        facts = set()
        for a in RADICES:
            for b in RADICES:
                if L % (a * b) == 0 and (L // (a * b)) in RADICES and a <= b <= L // (a * b):
                    facts.add((a, b, L // (a * b)))
        facts = sorted(facts, key=lambda f: f[2] / f[0])[:2]         # the two most balanced
        for f in facts:
            for perm in set(itertools.permutations(f)):
                if fits(*perm):
                    cands.append(perm)
    print("%d candidate kernels" % len(cands), file=sys.stderr)
    nproc = 8
    chunks = [cands[i::nproc] for i in range(nproc)]
    spills = {}
    with ThreadPoolExecutor(nproc) as ex:
        for r in ex.map(lambda t: compile_triples(*t), enumerate(chunks)):
            spills.update(r)
    best = {}
    for (a, b, c), s in sorted(spills.items()):
        L = a * b * c
        if s > MAX_SPILL:
            continue
        key = (s, -tile(a, b, c), abs(a - c))
        if L not in best or key < best[L][3]:
            best[L] = (a, b, c, key)
    with open(os.path.join(CSRC, "r3_menu.inc"), "w") as f:
        f.write("/* generated by tools/gen_r3_menu.py -- X(L, R1, R2, R3): three-stage rows kernel of length L */\n")
        for L in sorted(best):
            f.write("X(%d, %d, %d, %d)\n" % ((L,) + best[L][:3]))
    print("%d lengths" % len(best), file=sys.stderr)
    # strided / transposed forms (pass3t_kernel): L <= 2048, their own best factor order; lengths that have a
    # two-stage kernel never reach this one
    tc = [c for c in cands if fits_t(*c)]
    if 2048 not in have2:
        tc.append((8, 16, 16))
    chunks = [tc[i::nproc] for i in range(nproc)]
    spills = {}
    with ThreadPoolExecutor(nproc) as ex:
        for r in ex.map(lambda t: compile_triples(t[0], t[1], True), enumerate(chunks)):
            spills.update(r)
    best = {}
    for (a, b, c), sp in sorted(spills.items()):
        L = a * b * c
        if sp > MAX_SPILL:
            continue
        key = (sp, abs(a - c))
        if L not in best or key < best[L][3]:
            best[L] = (a, b, c, key)
    with open(os.path.join(CSRC, "r3t_menu.inc"), "w") as f:
        f.write("/* generated by tools/gen_r3_menu.py -- X(L, R1, R2, R3): three-stage kernel of length L for\n"
                "   column passes and transposed last passes (pass3t_kernel) */\n")
        for L in sorted(best):
            f.write("X(%d, %d, %d, %d)\n" % ((L,) + best[L][:3]))
    print("%d strided lengths" % len(best), file=sys.stderr)


if __name__ == "__main__":
    main()
