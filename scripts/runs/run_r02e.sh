#!/bin/bash
source scripts/gpu_steps.sh
step 600 pytest_r02e.log python -m pytest tests/test_gpu_parity.py tests/test_bench_launcher.py -m gpu -q
step 400 bench_r02e.json python bench.py --steps 5 --warmup 2
for wl in stencil3d_32x32_c st:64:64:c:24:24:4 st:32:32:z:48:48:4 st:64:64:z:24:24:4 st:32:64:c:32:32:4; do
  for c in 0 1; do
    step 200 clamp_${wl//:/_}_$c.txt env TFQMRGPU_CLAMP=$c python scripts/bench_multiply.py $wl 10
  done
done
grep -E "passed|failed|FAILED" gpurun_out/pytest_r02e.log | tail -8
for f in gpurun_out/clamp_*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter" $f | cut -c1-160; done
