#!/bin/bash
source scripts/gpu_steps.sh
step 900 pytest_ilvf.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash_mode.py tests/test_gpu_configs.py tests/test_gpu_operator.py -m gpu -q -x
grep -E "^FAILED|passed|failed|Error" gpurun_out/pytest_ilvf.log | tail -6
for wl in stencil3d_32x32_c st:16:32:c:64:64:8 st:16:64:c:48:48:4 st:32:64:c:32:32:4 st:64:64:c:24:24:4; do
  t=$(echo $wl | tr ':' '_')
  step 300 ilvf_${t}_1.txt python scripts/bench_multiply.py $wl 10
  step 300 ilvf_${t}_3.txt env TFQMRGPU_ILV=3 python scripts/bench_multiply.py $wl 10
done
for f in gpurun_out/ilvf_*.txt; do echo "== $f"; grep -E "spmm|per iter|status" $f | cut -c1-120; done
