#!/bin/bash
# round-3 shape survey: all 15 block shapes x z, c (scripts/shape_survey.sh), to compare with profiles/r01_shape_survey.txt
source scripts/gpu_steps.sh
timeout 1100 bash scripts/shape_survey.sh > gpurun_out/r03_shape_survey.txt 2>&1
wc -l gpurun_out/r03_shape_survey.txt
