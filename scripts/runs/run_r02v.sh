#!/bin/bash
source scripts/gpu_steps.sh
for g in 4 2 3 6 1; do
step 300 orderg_$g.txt env TFQMRGPU_ORDER_G=$g python scripts/bench_multiply.py fd2d_16x16_z 10
done
for f in gpurun_out/orderg_*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter" $f | cut -c1-70; done
