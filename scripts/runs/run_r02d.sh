#!/bin/bash
source scripts/gpu_steps.sh
step 900 pytest_r02d.log python -m pytest tests -m gpu -q
step 600 parity_report3.txt python tests/parity_report.py
step 400 bench_r02d.json python bench.py --steps 5 --warmup 2
grep -E "passed|failed|FAILED" gpurun_out/pytest_r02d.log | tail -15
