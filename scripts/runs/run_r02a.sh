#!/bin/bash
source scripts/gpu_steps.sh
step 900 pytest_r02a.log python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_hash_mode.py::test_hash_mode_takes_the_oracles_trajectory_z
step 600 pytest_r02a_hash.log python -m pytest tests/test_gpu_hash_mode.py -m gpu -q
step 600 parity_report.txt python tests/parity_report.py
step 300 mfma_rate.txt scripts/bin/mfma_rate
