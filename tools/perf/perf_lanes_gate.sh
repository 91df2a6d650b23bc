for w in 2d mixed; do for l in 1 2 1 2; do
FFTW_AMD_LANES=$l python bench.py --workload $w --no-legs --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$w lanes=$l', 'ms_per_step %.3f min %.3f' % (d['ms_per_step'], d['ms_per_step_min']), d['roofline']['per_launch']['concurrent_lanes'])
"
done; done
