#!/usr/bin/env python
"""bench.py -- headline benchmark of the FFT executor on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2c|r2c|mixed|2d|dct2|dct2-2d|slab2d]

A "step" is one fftw_execute of the whole batch of synthetic input that is
already resident in HBM.  At N=1 the default workload is BASELINE.json
configs[1]: 1-D complex double N = 2^20, howmany = 4096 (fftw_plan_many_dft),
forward, out of place.  The metric is the reference's own
(fftw/libbench2/mflops.c:21-28): GFLOPS = 5 N log2 N * howmany / t.

With --gpus N > 1 the driver launches one process per GPU
(torch.distributed.run); whole batch elements are sharded across ranks, every
rank transforms the same per-GPU batch (weak scaling) and nothing is exchanged
inside the timed region.  `--gather` adds the RCCL all-gather of the outputs
(reported separately: it is xGMI-bound).

Rank 0 prints ONE JSON line.  It carries `roofline` for the dominant kernel
(HIP events around every launch, on the stream the kernels run on) and, at
N=1, `cpu_baseline`: the CPU oracle timed on one host core on a bounded sample
of the same workload.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def workload(name, batch_override):
    """(rank dims, howmany, kind, flops per transform, algorithmic bytes per transform)"""
    if name in ("c2c", "c2c-bwd-inplace"):
        n = [1 << 20]
        b = 4096
        kind = "c2c"
    elif name in ("r2c", "c2r"):
        n = [1 << 22]
        b = 1024
        kind = "r2c"
    elif name == "mixed":
        n = [3 * 5 * 7 * 11 * 13 * 1024]
        b = 256           # 2048 does not fit one GPU (504 GB per array): sub-batch of 256
        kind = "c2c"
    elif name == "2d":
        n = [4096, 4096]
        b = 64            # per-GPU share of 512 images on 8 GPUs
        kind = "c2c"
    elif name == "dct2":
        n = [1 << 20]
        b = 2048
        kind = "r2r"      # REDFT10 (DCT-II), SURVEY.md 8(f) row 3
    elif name == "dct2-2d":
        n = [4096, 4096]
        b = 32
        kind = "r2r"
    else:
        raise SystemExit("unknown workload " + name)
    if batch_override:
        b = batch_override
    size = 1
    for v in n:
        size *= v
    if kind == "c2c":
        flops = 5.0 * size * math.log2(size)
        abytes = 32.0 * size
    elif kind == "r2r":
        flops = 2.5 * size * math.log2(size)      # reference libbench2/mflops.c:25-28
        abytes = 16.0 * size                      # n reals in, n reals out
    else:
        flops = 2.5 * size * math.log2(size)
        abytes = 8.0 * size + 16.0 * (size // n[-1]) * (n[-1] // 2 + 1)
        if name == "c2r":
            kind = "c2r"
    return n, b, kind, flops, abytes


def pmc_traffic(name, alg_bytes_per_step):
    """(HBM-side bytes per step of the whole plan, source) from the committed PMC passes of THIS round's
    build (profiles/r03_traffic.json: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of
    this same bench command, FETCH_SIZE doubled as the MI355X guide prescribes for 16-B-per-lane streaming
    reads; the counters sit on the L2's fabric side, so Infinity-Cache hits are included).  Counters cannot
    be read inside this process, so the figure is static and `traffic_source` says so.  (None, None) when
    the workload has no committed measurement."""
    path = os.path.join(ROOT, "profiles", "r03_traffic.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        prof = json.load(f)
    w = prof.get("workloads", {}).get(name)
    if not w:
        return None, None
    return w["traffic_over_algorithmic"] * alg_bytes_per_step, (
        "profiles/r03_traffic.json (static: rocprofv3 --pmc passes of the same command; %.3f x the algorithmic bytes)"
        % w["traffic_over_algorithmic"])


# reference numbers measured by the survey on this container's CPU: the
# port below is NOT FFTW's speed
CPU_NOTE = ("the oracle is a plain-C restatement (kind 'port'); the reference itself measured on the survey "
            "container at n=2^20: 3.16 GFLOPS scalar build, 8.27 GFLOPS AVX+FFTW_MEASURE, one core (BASELINE.md section 2)")

_CPU_WORKER = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
core, kind, seconds = int(sys.argv[2]), sys.argv[3], float(sys.argv[4])
n = [int(v) for v in sys.argv[5].split("x")]
try:
    os.sched_setaffinity(0, {core})
except Exception:
    pass
import numpy as np
from util import oracle_dft, oracle_r2c, oracle_r2r
rng = np.random.default_rng(1 + core)
size = int(np.prod(n))
if kind == "c2c":
    x = (rng.random(size) - 0.5) + 1j * (rng.random(size) - 0.5)
    run = lambda: oracle_dft(x, tuple(n), 1)
elif kind == "r2r":
    x = rng.random(size) - 0.5
    run = lambda: oracle_r2r(x, list(n), [5] * len(n))
else:
    x = rng.random(size) - 0.5
    run = lambda: oracle_r2c(x, tuple(n), 1)
run()
print("ready", flush=True)
sys.stdin.readline()                      # start gun: every worker times the same window
t0 = time.perf_counter(); reps = 0
while True:
    run(); reps += 1
    dt = time.perf_counter() - t0
    if dt >= seconds:
        break
print("%d %.6f" % (reps, dt), flush=True)
"""


def cpu_baseline(n, kind, flops_per_transform, target_seconds=8.0, all_cores=True):
    """The oracle (kind "port": a plain-C restatement of the reference path) on the host cores of
    this box, on a bounded sample of the same workload: P pinned worker processes, each transforming
    its own share of the batch (the batch shards trivially, SURVEY.md 8d: "P independent processes,
    each planning howmany/P transforms"), all timing the same window.  Primary value: ONE core;
    `all_cores`: every core this process may use (capped at 16, the GPU box's CPU share)."""
    import subprocess
    import tempfile
    try:
        cores = sorted(os.sched_getaffinity(0))
    except Exception:
        cores = list(range(os.cpu_count() or 1))
    size = 1
    for v in n:
        size *= v
    # every worker holds its input, output and the oracle's tables (about 6 arrays of the size): the GPU
    # box gives one command ~270 GiB of host memory, so 16 workers fit every BASELINE config
    cap = max(1, min(16, int((96 << 30) // (size * 16 * 6 + 1))))
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(_CPU_WORKER)
        script = f.name

    def measure(use):
        procs = [subprocess.Popen([sys.executable, script, ROOT, str(c), kind, str(target_seconds),
                                   "x".join(str(v) for v in n)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  text=True, env=dict(os.environ, OMP_NUM_THREADS="1")) for c in use]
        try:
            for pr in procs:
                if pr.stdout.readline().strip() != "ready":
                    raise RuntimeError("cpu baseline worker failed to start")
            for pr in procs:
                pr.stdin.write("go\n")
                pr.stdin.flush()
            reps, worst = 0, 0.0
            for pr in procs:
                r, dt = pr.stdout.readline().split()
                reps += int(r)
                worst = max(worst, float(dt))
            for pr in procs:
                pr.wait(timeout=30)
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        return reps, worst

    try:
        reps1, dt1 = measure(cores[:1])
        out = {
            "value": flops_per_transform * reps1 / dt1 / 1e9, "unit": "GFLOPS", "cores": 1, "kind": "port",
            "sample": "%d transforms of n=%s (oracle/fftw_oracle.c, 1 pinned process, %.1f s)" % (
                reps1, "x".join(str(v) for v in n), dt1),
            "note": CPU_NOTE,
        }
        use = cores[:cap]
        if all_cores and len(use) > 1:
            repsP, dtP = measure(use)
            out["all_cores"] = {
                "value": flops_per_transform * repsP / dtP / 1e9, "unit": "GFLOPS", "cores": len(use), "kind": "port",
                "sample": "%d transforms of n=%s over %d pinned processes, one share of the batch each, %.1f s" % (
                    repsP, "x".join(str(v) for v in n), len(use), dtP),
            }
    finally:
        os.unlink(script)
    return out


def run_workload(name, batch_override, steps, warmup, torch, fa, dist, world, rank, dev, want_cpu, cpu_seconds=8.0):
    """One BASELINE config on this rank's GPU: plan, W warm-ups, K timed executions between
    barriers (wall clock, max over ranks) plus a HIP-event pair around every single step for
    min / median, the per-launch roofline of the dominant kernel, and the CPU baseline."""
    n, b, kind, flops1, abytes1 = workload(name, batch_override)
    size = 1
    for v in n:
        size *= v
    # shrink the batch if this GPU cannot hold input + output + scratch
    free, _ = torch.cuda.mem_get_info()
    per = (32 if kind == "c2c" else 8 + 17) * size
    while b > 1 and b * per + (1 << 30) > free:
        b //= 2

    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    check = None
    variant = "forward, out-of-place"
    if kind == "c2c":
        x = torch.empty((b, size), dtype=torch.complex128, device=dev)
        xr = torch.view_as_real(x)
        step_rows = max(1, (1 << 28) // size)
        for r0 in range(0, b, step_rows):        # uniform [-0.5, 0.5), filled in slabs
            sl = xr[r0:r0 + step_rows]
            sl.copy_(torch.rand(sl.shape, dtype=torch.float64, device=dev, generator=gen) - 0.5)
        if name == "c2c-bwd-inplace":
            # "also backward + in-place variants" of configs[1]
            variant = "backward, in-place"
            y = x
            plan = fa.plan_many_dft(len(n), n, b, x, None, 1, size, x, None, 1, size, fa.BACKWARD, fa.ESTIMATE)
        else:
            y = torch.empty_like(x)
            plan = fa.plan_many_dft(len(n), n, b, x, None, 1, size, y, None, 1, size, fa.FORWARD, fa.ESTIMATE)
    elif kind == "r2r":
        x = torch.rand((b, size), dtype=torch.float64, device=dev, generator=gen) - 0.5
        y = torch.empty_like(x)
        plan = fa.plan_many_r2r(len(n), n, b, x, None, 1, size, y, None, 1, size,
                                [fa.REDFT10] * len(n), fa.ESTIMATE)
    elif kind == "c2r":
        # "then c2r of the result (round-trip error check /N)": the spectrum comes from
        # this library's own r2c (untimed here; its parity is the r2c leg's and the GPU tests' business)
        variant = "backward (c2r of the r2c result), out-of-place"
        hs = size // n[-1] * (n[-1] // 2 + 1)
        x0 = torch.rand((b, size), dtype=torch.float64, device=dev, generator=gen) - 0.5
        x = torch.empty((b, hs), dtype=torch.complex128, device=dev)
        fwd = fa.plan_many_dft_r2c(len(n), n, b, x0, None, 1, size, x, None, 1, hs, fa.ESTIMATE)
        fwd.execute()
        torch.cuda.synchronize()
        del fwd
        keep = min(b, 8)
        ref = x0[:keep].clone()
        del x0
        torch.cuda.empty_cache()
        y = torch.empty((b, size), dtype=torch.float64, device=dev)
        plan = fa.plan_many_dft_c2r(len(n), n, b, x, None, 1, hs, y, None, 1, size, fa.ESTIMATE)

        def check():
            # c2r may overwrite its input (FFTW's contract without FFTW_PRESERVE_INPUT): checked on the FIRST run only
            got = y[:keep] / float(size)
            return float((got - ref).abs().max() / ref.abs().max())
    else:
        hs = size // n[-1] * (n[-1] // 2 + 1)
        x = torch.rand((b, size), dtype=torch.float64, device=dev, generator=gen) - 0.5
        y = torch.empty((b, hs), dtype=torch.complex128, device=dev)
        plan = fa.plan_many_dft_r2c(len(n), n, b, x, None, 1, size, y, None, 1, hs, fa.ESTIMATE)
    plan.set_stream(torch.cuda.current_stream().cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    roundtrip = None
    if check is not None:
        plan.execute()
        torch.cuda.synchronize()
        roundtrip = check()
    for _ in range(warmup):
        plan.execute()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record()                     # torch's current stream IS the stream the plan launches on
        plan.execute()
        e1.record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    total_transforms = b * world * steps
    gflops = flops1 * total_transforms / dt / 1e9
    alg_gbs = abytes1 * total_transforms / dt / 1e9

    # ---- roofline: achieved = ALGORITHMIC bytes / time -- every transform counted once,
    # read once + written once, whatever the number of trips the plan makes -- over the timed region of
    # this rank; the dominant kernel's own launches (HIP events around every launch, on the stream the
    # kernel is launched on) are reported beside it under per_launch
    prof = plan.execute_profiled()
    torch.cuda.synchronize()
    roof = None
    if prof:
        dom = max(prof, key=lambda t: t[1])
        st, ms, launches = dom
        avg_ms = ms / max(1, launches)
        units = min(plan.chunk, plan.batch)           # transforms one launch processes
        # a pass reads every element of its chunk once and writes it once
        # (r2r: every step of the REDFT10 plan moves n reals in and n reals out per transform,
        # the half-length complex passes included)
        bytes_per_launch = abytes1 * units
        kernel = "step%d kind=%d L=%d variant=%d" % (prof.index(dom), st.kind, st.L, st.variant)
        traffic, source = pmc_traffic(name, abytes1 * b)
        if plan.paired:
            # one launch = pass 2 of an earlier chunk + pass 1 of chunk c: both passes' bytes, except that
            # the first and the last launches hold one pass only
            nch = (plan.batch + plan.chunk - 1) // plan.chunk
            nl = max(1, launches)
            bytes_per_launch = 2.0 * abytes1 * units * nch / nl
            kernel = "pass1024_pair_kernel (pass 2 of an earlier chunk + pass 1 of chunk c in one launch)"
        pl_achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        whole = alg_gbs / world
        roof = {
            "bound": "hbm", "achieved": whole, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": whole / HBM_PEAK_GBS,
            "definition": "algorithmic bytes (%d per transform: input read once + output written once) x transforms / timed region" % int(abytes1),
            "traffic": traffic, "traffic_source": source,
            "per_launch": {
                "kernel": kernel, "achieved": pl_achieved, "frac": pl_achieved / HBM_PEAK_GBS,
                "avg_launch_ms": avg_ms, "launches_per_step": launches, "alg_bytes_per_launch": bytes_per_launch,
                "concurrent_lanes": plan.lanes,
                "note": "bytes the launch itself reads + writes (one pass over its chunk) / its HIP-event duration on its own "
                        "stream; with chunk lanes > 1 that many launches share the chip at any time, so the chip-level rate "
                        "is about lanes x this figure and the sum of steps_ms exceeds ms_per_step",
            },
            "steps_ms": [round(t[1], 4) for t in prof],
        }
    res = {
        "workload": name, "kind": kind, "n": n, "howmany": b, "size": size, "flops1": flops1, "abytes1": abytes1,
        "gflops": gflops, "alg_gbs": alg_gbs, "ms_per_step": dt / steps * 1e3,
        "ms_min": step_ms[0], "ms_median": step_ms[len(step_ms) // 2],
        "plan": plan.sprint().replace("\n", " "), "roofline": roof, "free": free, "variant": variant,
        "roundtrip_rel_err": roundtrip,
    }
    res["_y"] = y
    if want_cpu:
        res["cpu_baseline"] = cpu_baseline(n, kind, flops1, target_seconds=cpu_seconds)
    del plan, x
    return res


def run_slab(args, torch, fa, dist, world, rank, dev):
    """--workload slab2d: ONE 16384 x 16384 complex transform distributed in slabs over the
    ranks (fftw3_amd.slab): strong scaling, two all-to-all exchanges
    per transform (normal in, normal out)."""
    from fftw3_amd import slab
    n0 = n1 = args.batch or 16384
    alloc, ln0, s0, ln1, s1 = slab.local_size_2d_transposed(n0, n1, world, rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    x = torch.view_as_complex(torch.rand((max(1, alloc), 2), dtype=torch.float64, device=dev, generator=gen) - 0.5)
    y = torch.empty_like(x)
    plan = slab.plan_dft_2d(n0, n1, x, y, fa.FORWARD, fa.ESTIMATE, world=world, rank=rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        plan.execute()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.execute()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # time of the exchanges alone (same buffers, same splits)
    ex = [p for k, p in plan.stages if k == "a2a"]
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for e in ex:
            e.run()
    barrier()
    dte = time.perf_counter() - t0
    size = n0 * n1
    flops = 5.0 * size * math.log2(size)
    if rank == 0:
        sent = sum(sum(e.send_counts) - e.send_counts[rank] for e in ex) * 16
        print(json.dumps({
            "metric": "GFLOPS (5N*log2N), one 2D complex double transform %dx%d in slabs over the GPUs" % (n0, n1),
            "value": flops * args.steps / dt / 1e9, "unit": "GFLOPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "slab2d %dx%d c2c forward, normal in / normal out" % (n0, n1),
                       "local_plans": plan.num_local_plans(), "exchanges": plan.num_exchanges(),
                       "parallelism": "slab x%d" % world},
            "exchange": {"ms_per_step": dte / args.steps * 1e3, "bytes_sent_per_rank_per_step": sent,
                         "GBs_per_rank": sent * args.steps / dte / 1e9 if dte > 0 else None},
        }))
    if dist is not None:
        dist.destroy_process_group()


def spawn_ranks(nproc, argv, script=None, timeout=None):
    """One fresh child process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its
    environment (the contract of `python -m torch.distributed.run --nproc-per-node N`).
    The batch split the ranks then apply is the block rule of fftw/mpi/block.c:35-42.
    Rank 0's stdout (the one JSON line) is relayed; returns non-zero if any rank failed.
    Children are started with subprocess (never an exec of this process) and killed as a
    group of exact PIDs if one of them dies."""
    import socket
    import subprocess
    script = script or os.path.abspath(__file__)
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    procs = []
    for r in range(nproc):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    t_end = None if timeout is None else time.monotonic() + timeout
    alive = set(range(nproc))
    out0 = b""
    try:
        while alive:
            for r in sorted(alive):
                p = procs[r]
                try:
                    if r == 0:
                        o, _ = p.communicate(timeout=0.2)
                        out0 += o or b""
                    else:
                        p.wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
                alive.discard(r)
                if p.returncode != 0:
                    rc = rc or p.returncode or 1
                    sys.stderr.write("bench.py: rank %d exited with code %s\n" % (r, p.returncode))
            if rc and alive or (t_end is not None and time.monotonic() > t_end):
                rc = rc or 124
                break
    finally:
        for r in alive:
            procs[r].kill()
        for r in alive:
            try:
                o, _ = procs[r].communicate(timeout=10)
                if r == 0:
                    out0 += o or b""
            except Exception:
                pass
    # ONE JSON line on stdout; library chatter of rank 0 (e.g. "[Gloo] Rank 0 is connected ...") to stderr
    for line in out0.decode(errors="replace").splitlines():
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2c")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU howmany override")
    ap.add_argument("--gather", action="store_true", help="also time the RCCL all-gather of outputs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline workload only (no backward / r2c / c2r / mixed / 2d legs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.
        # It has not touched the GPU (no torch, no fftw3_amd yet) and never will.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import fftw3_amd as fa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # FFTW_AMD_BENCH_REHEARSE=1: rehearsal of the multi-rank path on a box with fewer GPUs
        # than ranks (ranks share cuda:0, gloo in the place of RCCL); never used for numbers
        rehearse = os.environ.get("FFTW_AMD_BENCH_REHEARSE") == "1"
        # the collective libraries announce themselves on fd 1 while they initialise ("[Gloo] Rank 0 is
        # connected ...", RCCL's "Hostname / Librccl path" banner): send fd 1 to stderr for that long, so that
        # the ONE JSON line stays alone on stdout whichever launcher started this rank
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearse:
                local_rank = local_rank % torch.cuda.device_count()
                torch.cuda.set_device(local_rank)
                dist.init_process_group("gloo")
            else:
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()                  # communicators are really built on first use
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    if fa.device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device: the executor has no CPU path")

    if args.workload == "slab2d":
        return run_slab(args, torch, fa, dist, world, rank, dev)

    res = run_workload(args.workload, args.batch, args.steps, args.warmup, torch, fa, dist, world, rank, dev,
                       want_cpu=(world == 1 and rank == 0 and not args.no_cpu_baseline))
    y = res.pop("_y")
    n, b, kind, size, free = res["n"], res["howmany"], res["kind"], res["size"], res["free"]

    gather = None
    if args.gather and dist is not None:
        from fftw3_amd.parallel import ShardedManyDft  # noqa: F401  (same collective as the helper)
        gb = min(b, max(1, (int(free * 0.35) // (16 * size)) // world))
        src = y[:gb].contiguous()
        dst = torch.empty((world * gb, y.shape[1]), dtype=y.dtype, device=dev)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(torch.view_as_real(dst), torch.view_as_real(src))
        dist.barrier()
        torch.cuda.synchronize()
        tg = time.perf_counter() - t0
        gather = {"seconds": tg, "bytes_received_per_rank": (world - 1) * src.numel() * 16,
                  "GBs_per_rank": (world - 1) * src.numel() * 16 / tg / 1e9}
        del src, dst
    del y
    torch.cuda.empty_cache()

    # ---- the other BASELINE configs, one GPU, same measurement (reported as legs, never as `value`)
    legs = []
    if world == 1 and args.workload == "c2c" and not args.no_legs and not args.batch:
        base_cfg = {"c2c-bwd-inplace": "configs[1], backward + in-place variant (SURVEY.md 8d row 2)",
                    "r2c": "configs[2]", "c2r": "configs[2], c2r of the r2c result (SURVEY.md 8d row 3)",
                    "mixed": "configs[3] (sub-batch 256 of 2048: 504 GB per array do not fit one GPU)",
                    "2d": "configs[4] (one GPU's share of 512 images on 8 GPUs)"}
        cpu_of = {}
        for name in os.environ.get("FFTW_AMD_BENCH_LEGS", "c2c-bwd-inplace,r2c,c2r,mixed,2d").split(","):
            # the backward / c2r variants run the same CPU code path as their forward twins: one CPU baseline each pair
            twin = {"c2c-bwd-inplace": "c2c", "c2r": "r2c"}.get(name)
            lr = run_workload(name, 0, max(3, args.steps // 2), 1, torch, fa, None, 1, 0, dev,
                              want_cpu=not args.no_cpu_baseline and twin is None, cpu_seconds=4.0)
            lr.pop("_y")
            torch.cuda.empty_cache()
            if twin is None:
                cpu_of[name] = lr.get("cpu_baseline")
            leg = {
                "workload": "%s n=%s howmany=%d, %s, FFTW_ESTIMATE" % (
                    lr["kind"], "x".join(str(v) for v in lr["n"]), lr["howmany"], lr["variant"]),
                "baseline_config": base_cfg[name],
                "value": lr["gflops"], "unit": "GFLOPS", "ms_per_step": lr["ms_per_step"],
                "ms_min": lr["ms_min"], "ms_median": lr["ms_median"], "algorithmic_GBs": lr["alg_gbs"],
                "plan": lr["plan"],
                "roofline": lr["roofline"],
                "cpu_baseline": lr.get("cpu_baseline") if twin is None else (
                    res.get("cpu_baseline") if twin == "c2c" else cpu_of.get(twin)),
            }
            if lr["roundtrip_rel_err"] is not None:
                leg["roundtrip_rel_err"] = lr["roundtrip_rel_err"]
                leg["roundtrip_check"] = "max |c2r(r2c(x)) / N - x| / max |x| over the first 8 transforms, bar 1e-10"
                if not (lr["roundtrip_rel_err"] <= 1e-10):
                    raise SystemExit("bench.py: c2r round trip off by %g" % lr["roundtrip_rel_err"])
            legs.append(leg)

    if rank == 0:
        out = {
            "metric": "GFLOPS (5N*log2N) + achieved HBM GB/s, 1D complex double N=2^20 batch=4096"
            if args.workload == "c2c" else "GFLOPS (reference mflops formula), workload " + args.workload,
            "value": res["gflops"], "unit": "GFLOPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
            "ms_per_step_min": res["ms_min"], "ms_per_step_median": res["ms_median"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s n=%s howmany=%d per GPU, %s, FFTW_ESTIMATE" % (
                kind, "x".join(str(v) for v in n), b, res["variant"]),
                "algorithmic_GBs": res["alg_gbs"], "parallelism": "batch-sharded x%d" % world,
                "plan": res["plan"]},
            "roofline": res["roofline"],
        }
        if legs:
            out["config"]["legs"] = legs
        if gather:
            out["all_gather"] = gather
        if "cpu_baseline" in res:
            out["cpu_baseline"] = res["cpu_baseline"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
