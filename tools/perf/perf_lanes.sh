#!/bin/bash
# chunk lanes x chunk size on the headline workload (and the legs): ms per step
out=gpurun_out/r03_lanes.txt
: > $out
for w in c2c r2c mixed 2d; do
  for cfg in "1 268435456" "2 33554432" "2 67108864" "2 100663296" "2 134217728" "2 201326592" "3 67108864" "3 100663296" "4 67108864"; do
    set -- $cfg
    echo "== $w lanes=$1 chunk=$2" >> $out
    FFTW_AMD_LANES=$1 FFTW_AMD_CHUNK_BYTES=$2 python bench.py --workload $w --no-legs --no-cpu-baseline --steps 10 --warmup 2 2>>$out | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms_per_step %.3f min %.3f value %.0f frac %.4f' % (d['ms_per_step'], d['ms_per_step_min'], d['value'], d['roofline']['frac']))
print('   steps_ms', d['roofline']['steps_ms'], 'plan', d['config']['plan'][:150])
" >> $out 2>&1
  done
done
