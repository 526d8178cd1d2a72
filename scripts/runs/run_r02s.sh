#!/bin/bash
source scripts/gpu_steps.sh
for r in 1 2; do
step 300 hq_late_$r.txt python scripts/bench_multiply.py fd2d_16x16_z 10
step 300 hq_early_$r.txt env TFQMRGPU_LIB=$PWD/scripts/bin/abl/libtfQMRgpu.so python scripts/bench_multiply.py fd2d_16x16_z 10
done
for f in gpurun_out/hq_*.txt; do echo "== $f"; grep -E "spmm|per iter" $f | cut -c1-70; done
