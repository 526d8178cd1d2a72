#!/bin/bash
source scripts/gpu_steps.sh
for wl in fd2d_16x16_z stencil3d_32x32_c; do
for v in base ilp memcl; do
  if [ $v = base ]; then step 300 sched_${wl}_$v.txt python scripts/bench_multiply.py $wl 5
  else step 300 sched_${wl}_$v.txt env TFQMRGPU_LIB=$PWD/scripts/bin/$v/libtfQMRgpu.so python scripts/bench_multiply.py $wl 5; fi
done; done
for f in gpurun_out/sched_*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter|status" $f | cut -c1-90; done
