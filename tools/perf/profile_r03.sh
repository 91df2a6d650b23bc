#!/bin/bash
# round-3 evidence (run on the GPU box): for the headline and every leg, the rocprofv3 kernel statistics of the
# bench command itself and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the same command.
#   rocprofv3 ... -- python3 bench.py --workload W ...     (the program directly after --)
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=$R/gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
cd $R
for W in ${WORKLOADS:-c2c r2c c2r mixed 2d}; do
  echo "== $W: kernel stats" 
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- python3 bench.py --workload $W --steps 5 --warmup 2 --no-legs --no-cpu-baseline > $O/r03_bench_${W}_under_rocprof.json 2> $O/stats_$W.err || { echo "stats run failed for $W"; tail -5 $O/stats_$W.err; }
  S=$(find $O/stats_$W -name "*kernel_stats.csv" | head -1)
  python3 - "$S" $O/r03_${W}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(rows[0])
    for r in rows[1:]:
        if "at::" in r[0] or "rocprim" in r[0] or "hipcub" in r[0]:
            continue                      # torch's RNG / fill kernels of the input set-up
        w.writerow([r[0][:160]] + r[1:])
PY
  head -4 $O/r03_${W}_kernel_stats.csv | cut -c1-200
  if [ "$W" != "c2r" ]; then
    for C in FETCH_SIZE WRITE_SIZE; do
      echo "== $W: pmc $C"
      rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${C}_$W -- python3 bench.py --workload $W --steps 1 --warmup 1 --no-legs --no-cpu-baseline > $O/pmc_${C}_$W.json 2> $O/pmc_${C}_$W.err || { echo "pmc run failed"; tail -5 $O/pmc_${C}_$W.err; }
    done
  fi
  rm -rf $O/stats_$W/*/*_agent_info.csv
done
python3 tools/perf/pmc_summary_r03.py $O $O/r03_traffic.json
ls -la $O | head -40
