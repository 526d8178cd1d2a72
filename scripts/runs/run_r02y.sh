#!/bin/bash
source scripts/gpu_steps.sh
for m in 0 4 8 16 32 64 128 256; do
step 300 orderm_$m.txt env TFQMRGPU_ORDER_M=$m python scripts/bench_multiply.py fd2d_16x16_z 5
done
for f in gpurun_out/orderm_*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter|status" $f | cut -c1-66; done
