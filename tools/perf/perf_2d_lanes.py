"""cfg5 (2-D c2c 4096 x 4096, batch 64): serial launches (the default: one transform's scratch, 256 MiB, exceeds the
128 MiB chunk budget, so one lane) against two chunk lanes of one transform each (FFTW_AMD_CHUNK_BYTES=256 MiB)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
n0 = n1 = 4096
hm = 64
x = torch.view_as_complex(torch.rand((hm * n0 * n1, 2), dtype=torch.float64, device="cuda") - 0.5)
y = torch.zeros_like(x)
for name, env in (("default", {}), ("chunk 256 MiB, 2 lanes", {"FFTW_AMD_CHUNK_BYTES": str(256 << 20), "FFTW_AMD_LANES": "2"}),
                  ("chunk 256 MiB, 3 lanes", {"FFTW_AMD_CHUNK_BYTES": str(256 << 20), "FFTW_AMD_LANES": "3"}),
                  ("chunk 512 MiB, 2 lanes", {"FFTW_AMD_CHUNK_BYTES": str(512 << 20), "FFTW_AMD_LANES": "2"})):
    for k in ("FFTW_AMD_CHUNK_BYTES", "FFTW_AMD_LANES"): os.environ.pop(k, None)
    os.environ.update(env)
    p = fa.plan_many_dft(2, [n0, n1], hm, x, None, 1, n0 * n1, y, None, 1, n0 * n1, fa.FORWARD)
    p.execute(); p.sync()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p.execute(); p.sync(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print("%-26s lanes %d chunk %d  %7.3f ms  %5.1f %%" % (name, p.lanes, p.chunk, t * 1e3, 100 * 32.0 * n0 * n1 * hm / t / 8e12), flush=True)
    del p
