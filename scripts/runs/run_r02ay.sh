#!/bin/bash
source scripts/gpu_steps.sh
for r in 1 2; do
step 300 hyb_base_$r.txt python scripts/bench_multiply.py fd2d_16x16_z 5
step 300 hyb_hyb_$r.txt env TFQMRGPU_LIB=$PWD/scripts/bin/hyb/libtfQMRgpu.so python scripts/bench_multiply.py fd2d_16x16_z 5
done
for f in gpurun_out/hyb_*.txt; do echo "== $f"; grep -E "spmm|per iter|status" $f | cut -c1-80; done
