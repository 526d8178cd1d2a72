#!/bin/bash
# the BASELINE configurations other than the bench line, for the record (profiles/r01_configs.txt)
echo "## config 1: reference plan file through the reference-style driver (bench_tfqmrgpu multi, f and z, 20 repetitions x 5 samples)"
python -m tfqmrgpu_amd.bench_tfqmrgpu multi tests/golden/plan_unordered.14-287-16.gz f 20 5 2>&1 | grep -v amdgpu.ids | tail -4
python -m tfqmrgpu_amd.bench_tfqmrgpu multi tests/golden/plan_unordered.14-287-16.gz z 20 5 2>&1 | grep -v amdgpu.ids | tail -4
echo "## config 1 through the compiled driver (tfqmrgpu_amd/lib/bench_tfqmrgpu multi, f and z)"
tfqmrgpu_amd/lib/bench_tfqmrgpu multi tests/golden/plan_unordered.14-287-16.gz f 20 5 2>&1 | grep -v amdgpu.ids | tail -4
tfqmrgpu_amd/lib/bench_tfqmrgpu multi tests/golden/plan_unordered.14-287-16.gz z 20 5 2>&1 | grep -v amdgpu.ids | tail -4
echo "## config 2 golden instance through the reference-style driver (bench_tfqmrgpu tfQMR)"
python -m tfqmrgpu_amd.bench_tfqmrgpu tfQMR tests/golden/fd_16x16_small.xml z 3 2000 2>&1 | grep -v amdgpu.ids | tail -4
for wl in stencil3d_32x32_c stencil2d_8x8_z; do
  echo "## bench.py --workload $wl"
  python bench.py --workload $wl --steps 20 --warmup 10 2>gpurun_out/configs_$wl.err | tail -1
done
echo "## config 4, the shard of one GPU: bench.py --workload st:16:16:z:128:128:32 (32 of the 256 block columns)"
python bench.py --workload st:16:16:z:128:128:32 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-multiply 2>gpurun_out/configs_cfg4.err | tail -1
echo "## bench.py --workload st:16:16:c:96:96:16 (16x16 complex<float>, the default shape and precision of the reference's bench multi)"
python bench.py --workload st:16:16:c:96:96:16 --steps 20 --warmup 10 --no-cpu-baseline --no-hbm-multiply 2>gpurun_out/configs_c16.err | tail -1
echo "## config 5 at the size SURVEY 8d restates it: bench.py --workload st:8:8:z:512:512:8 (262 144 block rows, 2.1 M X blocks, S = 2.15 GB)"
python bench.py --workload st:8:8:z:512:512:8 --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-multiply 2>gpurun_out/configs_c5big.err | tail -1
