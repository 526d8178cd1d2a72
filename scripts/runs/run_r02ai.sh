#!/bin/bash
source scripts/gpu_steps.sh
step 600 pytest_onecol.log python -m pytest tests/test_gpu_hash_mode.py -m gpu -q -k "onecol"
tail -15 gpurun_out/pytest_onecol.log
step 600 parity_onecol.txt python tests/parity_report.py
grep "onecol" gpurun_out/parity_onecol.txt
