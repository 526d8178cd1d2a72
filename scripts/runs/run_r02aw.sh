#!/bin/bash
source scripts/gpu_steps.sh
for wl in st:16:16:c:96:96:16 stencil2d_8x8_z; do
  t=$(echo $wl | tr ':' '_')
  for r in 1 2; do
  step 300 first_${t}_base_$r.txt python scripts/bench_multiply.py $wl 10
  step 300 first_${t}_all_$r.txt env TFQMRGPU_LIB=$PWD/scripts/bin/first3/libtfQMRgpu.so python scripts/bench_multiply.py $wl 10
  done
done
step 300 first_p2.txt python scripts/bench_multiply.py fd2d_16x16_z 5
for f in gpurun_out/first_*.txt; do echo "== $f"; grep -E "spmm|per iter" $f | cut -c1-80; done
