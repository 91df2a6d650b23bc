#!/bin/bash
# round-2 evidence: kernel statistics of the bench command + the two PMC passes (run on the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
FFTW_AMD_PAIR=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/perf/perf_traffic.py > $O/fetch.out 2> $O/fetch.err
FFTW_AMD_PAIR=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/perf/perf_traffic.py > $O/write.out 2> $O/write.err
find $O -name "*.csv" | head -20
F=$(find $O/fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/write -name "*counter_collection.csv" | head -1)
S=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python3 tools/perf/pmc_summary.py "$F" "$W" 16 $O/r02_traffic.json
head -6 "$S" | cut -c1-200 > $O/r02_c2c_kernel_stats_head.txt
# keep only the rows of this library's kernels (the torch RNG kernels have kilobyte-long names)
python3 - "$S" $O/r02_c2c_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(rows[0])
    for r in rows[1:]:
        w.writerow([r[0][:120]] + r[1:])
PY
python3 - "$F" $O/r02_pmc_fetch_size.csv "$W" $O/r02_pmc_write_size.csv <<'PY'
import csv, sys
for src, dst in ((sys.argv[1], sys.argv[2]), (sys.argv[3], sys.argv[4])):
    rows = list(csv.DictReader(open(src)))
    keep = [r for r in rows if "pass1024_kernel" in r["Kernel_Name"]]
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_ALL)
        w.writeheader()
        for r in keep[:96]:
            w.writerow(r)
PY
cat $O/r02_c2c_kernel_stats.csv | head -5 | cut -c1-220
cat $O/bench_under_rocprof.json | cut -c1-600
