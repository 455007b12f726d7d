#!/bin/bash
mkdir -p gpurun_out/r4v
timeout -k 10 1180 python -m pytest tests -q -m gpu --durations=30 > gpurun_out/r4v/gpu_suite_durations.txt 2>&1; rc=$?
echo "suite rc $rc"; grep -A34 "slowest" gpurun_out/r4v/gpu_suite_durations.txt | head -40; tail -3 gpurun_out/r4v/gpu_suite_durations.txt
