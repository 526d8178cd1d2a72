#!/bin/bash
source scripts/gpu_steps.sh
step 900 parity_report5.txt python tests/parity_report.py
grep -A40 "work vectors" gpurun_out/parity_report5.txt
