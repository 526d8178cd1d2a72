#!/bin/bash
# timing-only probes: 3 of 4 A fetches skipped (64), 3 of 4 X fetches skipped (128), both (192) -- what sharing operands between block products would buy
source scripts/gpu_steps.sh
export AB_MAXIT=30
timeout 800 python scripts/ab_fused.py fd2d_16x16_z tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_p64.so scripts/bin/libtfQMRgpu_p128.so scripts/bin/libtfQMRgpu_p192.so tfqmrgpu_amd/lib/libtfQMRgpu.so 2>&1 | grep -v amdgpu
