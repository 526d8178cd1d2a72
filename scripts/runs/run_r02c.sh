#!/bin/bash
source scripts/gpu_steps.sh
step 900 pytest_r02c.log python -m pytest tests -m gpu -q
step 600 parity_report2.txt python tests/parity_report.py
grep -E "passed|failed|FAILED" gpurun_out/pytest_r02c.log | tail -15
