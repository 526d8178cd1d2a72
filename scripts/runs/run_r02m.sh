#!/bin/bash
source scripts/gpu_steps.sh
step 300 hot2_c3.txt python scripts/hot_operand_probe.py stencil3d_32x32_c 20
step 300 hot2_64z.txt python scripts/hot_operand_probe.py st:64:64:z:24:24:4 20
for wl in stencil3d_32x32_c st:64:64:c:24:24:4 st:32:32:z:48:48:4 st:32:64:c:32:32:4; do
  step 200 sb_${wl//:/_}.txt python scripts/bench_multiply.py $wl 10
done
cat gpurun_out/hot2_*.txt | grep -E "^#|TFLOP"
for f in gpurun_out/sb_*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter" $f | cut -c1-170; done
