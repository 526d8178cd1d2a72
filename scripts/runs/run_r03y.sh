#!/bin/bash
# k_spmm_ilv16 forced to four waves per SIMD (launch bound 4 work groups per CU: 128 VGPRs, 24-104 bytes of scratch) against the product
source scripts/gpu_steps.sh
timeout 600 python scripts/ab_fused.py fd2d_16x16_z tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_w4.so tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_w4.so 2>&1 | grep -v amdgpu
