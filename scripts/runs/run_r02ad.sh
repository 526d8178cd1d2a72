#!/bin/bash
source scripts/gpu_steps.sh
for r in 1 2; do
step 300 prio_base_$r.txt python scripts/bench_multiply.py fd2d_16x16_z 5
step 300 prio_m2_$r.txt env TFQMRGPU_LIB=$PWD/scripts/bin/mprio2/libtfQMRgpu.so python scripts/bench_multiply.py fd2d_16x16_z 5
done
for f in gpurun_out/prio_base*.txt gpurun_out/prio_m2*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter" $f | cut -c1-70; done
