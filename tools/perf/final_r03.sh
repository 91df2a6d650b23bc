#!/bin/bash
# end-of-round evidence: 2-rank rehearsal of bench.py --gpus (gloo, both ranks on cuda:0), profiles of every workload, the default bench line
FFTW_AMD_BENCH_REHEARSE=1 python bench.py --gpus 2 --batch 64 --steps 3 --warmup 1 > gpurun_out/r03_rehearse2.json 2> gpurun_out/r03_rehearse2.err
tail -c 400 gpurun_out/r03_rehearse2.json; echo; tail -3 gpurun_out/r03_rehearse2.err
timeout -k 10 900 tools/perf/profile_r03.sh > gpurun_out/profile_r03.log 2>&1
grep "traffic / alg" gpurun_out/profile_r03.log
SECONDS=0; python bench.py > gpurun_out/r03_bench_3.json 2> gpurun_out/r03_bench_3.err
echo "bench wall ${SECONDS} s"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench_3.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["per_launch"]["avg_launch_ms"])
for l in d["config"]["legs"]:
    print(l["workload"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["traffic"], l["roofline"]["per_launch"]["avg_launch_ms"])
PY
