#!/bin/bash
source scripts/gpu_steps.sh
for pad in 0 4096 12288 45056 65536 262144 1052672; do
step 300 vpad_$pad.txt env TFQMRGPU_VPAD=$pad python scripts/bench_multiply.py fd2d_16x16_z 5
done
for f in gpurun_out/vpad_*.txt; do echo "== $f"; grep -E "xpay|v5_nrm|x_v6|spmm|per iter" $f | cut -c1-60; done
