#!/bin/bash
source scripts/gpu_steps.sh
step 600 r02_bench.json python bench.py --steps 5 --warmup 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r02_stats
step 400 rocprof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
step 1100 pytest_final.log python -m pytest tests -m gpu -q
grep -E "^FAILED|passed|failed" gpurun_out/pytest_final.log | tail -5
step 300 smoke.log python -c "import __graft_entry__ as g; g.smoke()"
tail -3 gpurun_out/smoke.log
