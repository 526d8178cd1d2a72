#!/bin/bash
source scripts/gpu_steps.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r03h_trace
step 300 r03h_trace.log rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r03h_trace -- python3 scripts/small_one.py
python3 scripts/trace_gaps.py gpurun_out/r03h_trace 60 > gpurun_out/r03h_gaps.txt 2>&1
rm -rf gpurun_out/r03h_trace
cat gpurun_out/r03h_gaps.txt
step 600 r03h_dev.log python -m pytest tests/test_gpu_parity.py -q -k "device_arrays"
tail -5 gpurun_out/r03h_dev.log
