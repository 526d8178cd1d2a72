#!/bin/bash
source scripts/gpu_steps.sh
step 900 parity_report7.txt python tests/parity_report.py
grep -A40 "hash shadow vector (the def" gpurun_out/parity_report7.txt | grep " z tol" | cut -c1-210
