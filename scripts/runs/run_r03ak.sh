#!/bin/bash
# the stopping decision in the last work group of k_decT<FINAL> / k_probe_col (one rank, unfolded plans): tests, then small-system and P2 timing
source scripts/gpu_steps.sh
step 900 pytest_r03ak.log python -m pytest tests/test_gpu_hash_mode.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_operator.py tests/test_gpu_ranks.py tests/test_gpu_mixed.py -q -x
tail -3 gpurun_out/pytest_r03ak.log
export TFQMRGPU_LIB=$PWD/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
for j in 0 1 0 1; do echo "TFQMRGPU_JOIN=$j"; TFQMRGPU_JOIN=$j python scripts/small_latency.py 2>&1 | grep -v amdgpu | tail -3; done
