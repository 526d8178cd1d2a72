#!/bin/bash
source scripts/gpu_steps.sh
step 300 aux_base.txt python scripts/bench_multiply.py fd2d_16x16_z 5
for v in aux_2_2 aux_18_18 aux_19_19 aux_17_17 aux_2_19 aux_18_2; do
step 300 $v.txt env TFQMRGPU_LIB=$PWD/scripts/bin/$v/libtfQMRgpu.so python scripts/bench_multiply.py fd2d_16x16_z 5
done
step 300 aux_base2.txt python scripts/bench_multiply.py fd2d_16x16_z 5
for f in gpurun_out/aux_*.txt; do echo "== $f"; grep -E "spmm|per iter|status" $f | cut -c1-70; done
