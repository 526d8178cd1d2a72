#!/bin/bash
# The record of a round, one gpurun call: the bench line (-> profiles/rNN_bench.json), rocprofv3 kernel stats of the same
# command (-> profiles/rNN_rocprofv3_kernel_stats.csv), the other BASELINE configurations (-> profiles/rNN_configs.txt),
# the whole GPU test suite and the smoke test.  usage: gpurun --timeout 1200 -- scripts/round_record.sh
source scripts/gpu_steps.sh
step 600 bench.json python bench.py --steps 20 --warmup 5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/stats
step 400 rocprof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
find gpurun_out/stats -name "*kernel_stats.csv" | head -2
step 1100 configs.txt bash scripts/run_configs.sh
step 1100 pytest_gpu.log python -m pytest tests -m gpu -q
grep -E "^FAILED|passed|failed" gpurun_out/pytest_gpu.log | tail -5
step 300 smoke.log python -c "import __graft_entry__ as g; g.smoke()"
tail -1 gpurun_out/smoke.log
