"""perf exploration: do Infinity-Cache hits and HBM misses add up? two load-only plans on two streams"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
dev = torch.device("cuda:0")
n = 1 << 20
fa.set_chunk_bytes(1 << 40)
P, R = 8, 64
xa = torch.randn(P, n, dtype=torch.complex128, device=dev); ya = torch.empty(8, n, dtype=torch.complex128, device=dev)
pa = fa.plan_guru64_dft([(n, 1, 1)], [(P, n, 0), (R, 0, 0)], xa, ya, -1)          # 512 transforms cycling through 128 MiB
B = 512
xb = torch.randn(B, n, dtype=torch.complex128, device=dev)
pb = fa.plan_guru64_dft([(n, 1, 1)], [(B, n, 0)], xb, ya, -1)                      # 512 transforms streaming 8 GiB
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
pa.set_stream(s1.cuda_stream); pb.set_stream(s2.cuda_stream)
def run(which):
    torch.cuda.synchronize(); t = time.perf_counter()
    if "a" in which: pa.execute()
    if "b" in which: pb.execute()
    torch.cuda.synchronize(); return time.perf_counter() - t
for w in ("a", "b", "ab", "a", "b", "ab"):
    dt = run(w)
    nx = (P * R if "a" in w else 0) + (B if "b" in w else 0)
    print("%-3s %.3f ms  %.2f us/xform-pass(both passes load-only)  aggregate read %.2f TB/s" % (w, dt * 1e3, dt / nx * 1e6, nx * 2 * 16 * n / dt / 1e12), flush=True)
