#!/bin/bash
# Does the FFT's arithmetic pull the clocks down?  Sample sclk / mclk / power while (a) the product runs the
# headline workload in a loop, (b) the pure-copy probe runs the same launch structure.
out=gpurun_out/r03_clocks.txt
: > $out
sample() {  # $1 = label, $2 = seconds
  for i in $(seq 1 $(( $2 * 2 ))); do
    echo "[$1] $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|socclk|Power' | tr -s ' ' | tr '\n' ';')" >> $out
    sleep 0.5
  done
}
echo "== idle" >> $out; sample idle 2
python - <<'PY' &
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, fftw3_amd as fa
dev = torch.device("cuda:0")
n, b = 1 << 20, 2048
x = torch.randn(b, n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
p = fa.plan_many_dft(1, [n], b, x, None, 1, n, y, None, 1, n, -1)
p.execute(); torch.cuda.synchronize()
t = time.time()
k = 0
while time.time() - t < 14:
    for _ in range(10): p.execute()
    torch.cuda.synchronize(); k += 10
print("fft loop: %.2f ms per 2048 transforms" % ((time.time() - t) / k * 1e3), flush=True)
PY
sleep 5
echo "== product FFT loop (c2c 2^20 x 2048)" >> $out; sample fft 7
wait
echo "== pure copy, same launch structure (mall_probe F)" >> $out
( for i in 1 2 3; do tests/micro/mall_probe F > /dev/null; done ) &
sleep 3; sample copy 6
wait
echo "== copy + FMA chain (mall_probe G)" >> $out
( for i in 1 2 3 4 5 6; do tests/micro/mall_probe G > /dev/null; done ) &
sleep 2; sample copyfma 6
wait
