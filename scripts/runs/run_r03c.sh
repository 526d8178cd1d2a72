#!/bin/bash
# round 3: exact vmcnt waits in k_spmm_ilv16 (no conditional load in the product loop): A/B against the r02 kernel
source scripts/gpu_steps.sh
step 600 r03c_ab.txt python scripts/ab_fused.py fd2d_16x16_z scripts/bin/libtfQMRgpu_r02.so default scripts/bin/libtfQMRgpu_tails.so scripts/bin/libtfQMRgpu_r02.so default
cat gpurun_out/r03c_ab.txt
step 900 r03c_pytest.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash_mode.py -q -x
