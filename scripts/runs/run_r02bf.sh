#!/bin/bash
source scripts/gpu_steps.sh
step 900 parity_report8.txt python tests/parity_report.py
step 1100 pytest_bf.log python -m pytest tests -m gpu -q
grep -E "^FAILED|passed|failed" gpurun_out/pytest_bf.log | tail -12
