#!/bin/bash
# round 3: column operations folded into the producers' tails for small systems
source scripts/gpu_steps.sh
step 300 r03g_small.txt python scripts/small_latency.py
cat gpurun_out/r03g_small.txt | grep -v amdgpu
export TFQMRGPU_LIB=$PWD/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
TFQMRGPU_FOLD_MAX=0 timeout 300 python scripts/small_latency.py 2>&1 | grep -v amdgpu | sed 's/^/nofold: /'
for fm in 0 100000; do
  echo "config 3, FOLD_MAX=$fm"; TFQMRGPU_FOLD_MAX=$fm timeout 300 python bench.py --workload stencil3d_32x32_c --steps 20 --warmup 10 --no-cpu-baseline --no-mixed 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline_iteration']['ms_per_iteration'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
done
unset TFQMRGPU_LIB
step 1100 r03g_pytest.log python -m pytest tests -m gpu -q -x
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r03g_pytest.log | tail -15
