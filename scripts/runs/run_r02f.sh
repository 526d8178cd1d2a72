#!/bin/bash
source scripts/gpu_steps.sh
for c in 0 1; do step 200 deep_$c.txt env TFQMRGPU_DEEP=$c python scripts/bench_multiply.py stencil3d_32x32_c 10; done
for c in 0 1; do step 200 deepb_$c.txt env TFQMRGPU_DEEP=$c python scripts/bench_multiply.py st:32:32:c:96:96:4 10; done
for f in gpurun_out/deep*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter|solve status" $f | cut -c1-170; done
