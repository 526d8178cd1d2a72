#!/bin/bash
source scripts/gpu_steps.sh
for c in 64 32 16 8; do
step 300 chk_$c.txt env TFQMRGPU_CHUNK_KIB=$c python scripts/bench_multiply.py fd2d_16x16_z 5
done
for f in gpurun_out/chk_*.txt; do echo "== $f"; grep -E "^multiply|spmm|xpay|v5_nrm|x_v6|dec35|per iter|status" $f | cut -c1-75; done
