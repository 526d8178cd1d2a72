#!/bin/bash
source scripts/gpu_steps.sh
step 600 parity_report4.txt python tests/parity_report.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step 400 rocprof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
step 900 pmc_p2.log bash scripts/pmc_collect.sh gpurun_out/r02_pmc fd2d_16x16_z
step 900 pmc_8x8.log bash scripts/pmc_collect.sh gpurun_out/r02_pmc8 stencil2d_8x8_z
python3 scripts/pmc_summary.py gpurun_out/r02_pmc > gpurun_out/r02_pmc_summary.json
python3 scripts/pmc_summary.py gpurun_out/r02_pmc8 > gpurun_out/r02_pmc_summary_8x8z.json
find gpurun_out/r02_stats -name "*kernel_stats.csv" | head -2
tail -2 gpurun_out/rocprof_bench.log | cut -c1-300
