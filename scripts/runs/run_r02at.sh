#!/bin/bash
source scripts/gpu_steps.sh
for wl in stencil3d_32x32_c st:16:32:c:64:64:8; do
  t=$(echo $wl | tr ':' '_')
  step 300 ilvg_${t}_1.txt python scripts/bench_multiply.py $wl 10
  step 300 ilvg_${t}_3.txt env TFQMRGPU_ILV=3 python scripts/bench_multiply.py $wl 10
done
for f in gpurun_out/ilvg_*.txt; do echo "== $f"; grep -E "spmm|per iter|status" $f | cut -c1-120; done
step 1100 pytest_at.log python -m pytest tests -m gpu -q
grep -E "^FAILED|passed|failed|Error" gpurun_out/pytest_at.log | tail -6
