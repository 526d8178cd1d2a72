#!/bin/bash
source scripts/gpu_steps.sh
step 900 r03e_mixed.log python -m pytest tests/test_gpu_mixed.py tests/test_bench_launcher.py tests/test_bench_binary.py -q
tail -40 gpurun_out/r03e_mixed.log
