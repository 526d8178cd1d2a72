#!/bin/bash
# samples GPU clock and power while the multiply / the solver runs (is the part power-limited under fp64 MFMA load?)
mkdir -p gpurun_out
rocm-smi --showclocks --showpower --showmaxpower 2>&1 | grep -v "^$" | head -40
echo "=== under load: stand-alone multiply x2000"
python scripts/bench_multiply.py fd2d_16x16_z 400 > gpurun_out/clock_probe_run.txt 2>&1 &
PID=$!
sleep 14
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo
  sleep 0.4
done
wait $PID
cat gpurun_out/clock_probe_run.txt | grep -E "^multiply|per iter"
