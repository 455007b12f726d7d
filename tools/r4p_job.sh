#!/bin/bash
mkdir -p gpurun_out/r4p
timeout -k 10 600 python -m pytest tests/test_c_example_gpu.py -x -q > gpurun_out/r4p/tests.txt 2>&1; echo rc $?; tail -15 gpurun_out/r4p/tests.txt
