#!/bin/bash
mkdir -p gpurun_out/r4q
timeout -k 10 1150 python -m pytest tests -x -q -m gpu --deselect tests/test_box_gpu.py --deselect tests/test_abi.py > gpurun_out/r4q/gpu_suite_rest.txt 2>&1; rc=$?
echo "suite (without test_box_gpu, which passed in r4o) rc $rc"; tail -8 gpurun_out/r4q/gpu_suite_rest.txt
