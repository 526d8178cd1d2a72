#!/bin/bash
# timing-only probes of k_spmm_ilv16 (TFQ_PROBE bits: 1 no products, 2 linear chunk order, 4 plain epilogue accesses, 8 no record reduction,
# 16 Y not stored, 32 no hash), 30 iterations each; results of these builds are wrong by construction
source scripts/gpu_steps.sh
export AB_MAXIT=30
libs="tfqmrgpu_amd/lib/libtfQMRgpu.so"
for b in 1 3 5 9 17 33 2 4 8 16 32 7; do libs="$libs scripts/bin/libtfQMRgpu_p$b.so"; done
timeout 800 python scripts/ab_fused.py fd2d_16x16_z $libs tfqmrgpu_amd/lib/libtfQMRgpu.so 2>&1 | grep -v amdgpu
