#!/bin/bash
# finer lanes x chunk sweep of the headline workload on one box (ms per step, min)
for cfg in "2 134217728" "3 67108864" "2 134217728" "3 67108864" "3 83886080" "2 117440512" "2 150994944" "3 50331648" "3 100663296" "2 100663296"; do
  set -- $cfg
  FFTW_AMD_LANES=$1 FFTW_AMD_CHUNK_BYTES=$2 python bench.py --workload c2c --no-legs --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('lanes=$1 chunk=%d MiB  ms_per_step %.3f min %.3f median %.3f' % ($2 >> 20, d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_median']))"
done
