#!/bin/bash
source scripts/gpu_steps.sh
step 600 pytest_xp.log python -m pytest tests/test_gpu_hash_mode.py tests/test_gpu_parity.py -m gpu -q -x
grep -E "^FAILED|passed|failed" gpurun_out/pytest_xp.log | tail -5
for r in 1 2; do
step 300 xp_new_$r.txt python scripts/bench_multiply.py fd2d_16x16_z 5
step 300 xp_head_$r.txt env TFQMRGPU_LIB=$PWD/scripts/bin/head/libtfQMRgpu.so python scripts/bench_multiply.py fd2d_16x16_z 5
done
for f in gpurun_out/xp_*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter" $f | cut -c1-70; done
