mkdir -p gpurun_out
for wl in r2c 2d mixed; do
  for mb in 64 128 256 512 1024; do
    echo "== $wl chunk ${mb} MiB"
    FFTW_AMD_CHUNK_BYTES=$((mb*1048576)) python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['roofline']['steps_ms'], d['config']['plan'][:60])"
  done
done
