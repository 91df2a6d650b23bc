"""Fold the rocprofv3 counter-collection CSVs of tools/perf/profile_r03.sh into profiles/r03_traffic.json.

    python tools/perf/pmc_summary_r03.py <prof_dir> <out.json>

One FETCH_SIZE pass and one WRITE_SIZE pass per workload, each over `python3 bench.py --workload W --steps 1
--warmup 1 --no-legs --no-cpu-baseline` = 3 executions of the plan (warm-up, timed step, per-launch profiled
execution).  FETCH_SIZE / WRITE_SIZE are in KB of 1024 B.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE tallies 128-byte requests at 64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact for
16-byte-per-lane streaming stores.  The counters sit on the L2's memory-side (fabric) requests: Infinity-Cache hits
are included, so this is traffic leaving the XCDs, an upper bound of the HBM traffic."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

EXECUTIONS = 3


def fold(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            if "at::" in name or "rocprim" in name or "hipcub" in name or "__amd_rocclr" in name:
                continue
            a = acc[name]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


def main():
    root, out = sys.argv[1], sys.argv[2]
    res = {}
    for w in ("c2c", "r2c", "mixed", "2d"):
        # newest first: gpurun merges the files of a later run beside those of an earlier one
        f = sorted(glob.glob(os.path.join(root, "FETCH_SIZE_" + w, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
        g = sorted(glob.glob(os.path.join(root, "WRITE_SIZE_" + w, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
        bj = os.path.join(root, "pmc_FETCH_SIZE_%s.json" % w)
        if not f or not g or not os.path.exists(bj):
            continue
        try:
            line = [l for l in open(bj) if l.lstrip().startswith("{")][-1]
            bench = json.loads(line)
        except Exception:
            continue
        alg_per_step = bench["config"]["algorithmic_GBs"] * 1e9 * bench["ms_per_step"] * 1e-3
        fetch, write = fold(f[0], "FETCH_SIZE"), fold(g[0], "WRITE_SIZE")
        kernels = {}
        tot = 0.0
        for name in sorted(set(fetch) | set(write)):
            rd = 2.0 * fetch[name][0] * 1024.0 if name in fetch else 0.0
            wr = write[name][0] * 1024.0 if name in write else 0.0
            n = max(fetch[name][1] if name in fetch else 0, write[name][1] if name in write else 0)
            kernels[name[:160]] = {"launches": n, "read_bytes_per_launch": rd / max(1, n), "write_bytes_per_launch": wr / max(1, n),
                                   "traffic_bytes_per_launch": (rd + wr) / max(1, n)}
            tot += rd + wr
        res[w] = {
            "bench_workload": bench["config"]["workload"], "plan": bench["config"]["plan"],
            "executions_profiled": EXECUTIONS,
            "algorithmic_bytes_per_step": alg_per_step,
            "traffic_bytes_per_step": tot / EXECUTIONS,
            "traffic_over_algorithmic": tot / EXECUTIONS / alg_per_step,
            "kernels": kernels,
        }
        print(w, "traffic / algorithmic = %.4f" % res[w]["traffic_over_algorithmic"])
    json.dump({"method": __doc__, "workloads": res}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
