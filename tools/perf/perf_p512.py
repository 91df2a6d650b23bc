"""exploration (not a test): per-step times of 1-D power-of-two sizes around 2^18"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fftw3_amd as fa
import ast
CASES = ast.literal_eval(os.environ.get("CASES", "[(17, None), (18, None), (19, None)]"))
for k, force in CASES:
    n = (1 << k) if k < 64 else k
    hm = max(1, (1 << 27) // n)
    if force:
        os.environ["FFTW_AMD_FORCE_LENS"] = force
    else:
        os.environ.pop("FFTW_AMD_FORCE_LENS", None)
    x = torch.view_as_complex(torch.rand((hm * n, 2), dtype=torch.float64, device="cuda") - 0.5)
    y = torch.zeros_like(x)
    p = fa.plan_many_dft(1, [n], hm, x, None, 1, n, y, None, 1, n, fa.FORWARD)
    for _ in range(2):
        p.execute()
    p.sync()
    r = p.execute_profiled()
    print(k, force, ["%.3f" % ms for (_, ms, _) in r], "sum %.3f" % sum(ms for (_, ms, _) in r))
    print("   ", p.sprint().replace("\n", " "))
    for (s, ms, c) in r:
        print("    L=%d is_l=%d os_l=%d dims n=%s is=%s os=%s tw=%s tile=%d var=%d flags=%x" % (
            s.L, s.is_l, s.os_l, list(s.dim_n[:s.ndims]), list(s.dim_is[:s.ndims]), list(s.dim_os[:s.ndims]), s.tw_n, s.tile, s.variant, s.flags))
