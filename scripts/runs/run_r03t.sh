#!/bin/bash
# config 3 (2 block columns x 1024 chunks): chunk length against the column kernels (lab build)
source scripts/gpu_steps.sh
L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
for kib in 16 64 128 256; do
  echo "stencil3d_32x32_c CHUNK_KIB=$kib"; TFQMRGPU_CHUNK_KIB=$kib timeout 300 python scripts/ab_fused.py stencil3d_32x32_c $L 2>&1 | grep -v amdgpu
done
step 300 r03t_launcher.log python -m pytest tests/test_bench_launcher.py -q
tail -3 gpurun_out/r03t_launcher.log
