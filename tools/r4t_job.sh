#!/bin/bash
mkdir -p gpurun_out/r4t
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/r4t/gpu_suite.txt 2>&1; rc=$?
echo "suite rc $rc"; tail -6 gpurun_out/r4t/gpu_suite.txt
