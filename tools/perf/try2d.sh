for fl in "" "32,128" "128,32" "16,256" "256,16" "64,64"; do
  echo "== FORCE_LENS=$fl"
  FFTW_AMD_FORCE_LENS=$fl python bench.py --workload 2d --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['roofline']['steps_ms'], d['config']['plan'][30:400])"
done
