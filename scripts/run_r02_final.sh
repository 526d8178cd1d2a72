#!/bin/bash
# the round's record: bench line, rocprofv3 kernel stats of the same command, the other BASELINE configurations
source scripts/gpu_steps.sh
step 600 r02_bench.json python bench.py --steps 5 --warmup 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r02_stats
step 400 rocprof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
find gpurun_out/r02_stats -name "*kernel_stats.csv" | head -2
step 1100 r02_configs.txt bash scripts/run_configs.sh
step 300 r02_c16.json python bench.py --workload st:16:16:c:96:96:16 --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-multiply
tail -c 300 gpurun_out/r02_bench.json
grep -c "^{" gpurun_out/r02_configs.txt
