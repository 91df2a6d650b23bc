"""1-D c2c n = L1 x L2 with BOTH lengths in 1025 ... 2048 in two trips (both passes on the 512-item strided /
transposed kernels, 8 ... 15 sequences per tile) against the three-trip plan: ms per ~4 GiB batch, whole % of the
8 TB/s roofline, ms per GiB of traffic (the unit of the planner's split costs), error against torch.fft"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
buf = torch.view_as_complex(torch.rand(((4 << 30) // 16, 2), dtype=torch.float64, device="cuda") - 0.5)
out = torch.zeros_like(buf)
pairs = [(2048, 2048), (2048, 1920), (1920, 2048), (2000, 2000), (2048, 1536), (1536, 2048), (1600, 1600), (1280, 2048), (2048, 1280),
         (1440, 1440), (1200, 1200), (1080, 1920), (1920, 1080)]
for L1, L2 in pairs:
    n = L1 * L2
    hm = buf.numel() // n
    x, y = buf[:hm * n], out[:hm * n]
    res = []
    for force in ("", "%d,%d" % (L1, L2)):
        if force: os.environ["FFTW_AMD_FORCE_LENS"] = force
        else: os.environ.pop("FFTW_AMD_FORCE_LENS", None)
        p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
        p.execute(); p.sync()
        ref = torch.fft.fft(x[:n])
        err = float((y[:n] - ref).abs().max() / ref.abs().max())
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
        res.append((min(ts), err, len(p.steps()), [s.L for s in p.steps()]))
        del p
    os.environ.pop("FFTW_AMD_FORCE_LENS", None)
    gib = 32.0 * n * hm / 2 ** 30
    (t3, e3, k3, l3), (t2, e2, k2, l2) = res
    print("%5d x %-5d default %d trips %-18s %7.3f ms %5.1f %% (%.3f ms/GiB) | forced %7.3f ms %5.1f %% (%.3f ms/GiB) err %.1e" %
          (L1, L2, k3, l3, t3 * 1e3, 100 * 32.0 * n * hm / t3 / 8e12, t3 * 1e3 / gib, t2 * 1e3, 100 * 32.0 * n * hm / t2 / 8e12, t2 * 1e3 / gib, max(e2, e3)), flush=True)
