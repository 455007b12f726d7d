#!/bin/bash
mkdir -p gpurun_out/r4u
timeout -k 10 1150 python -m pytest tests -x -q -m gpu --deselect tests/test_beams_gpu.py --deselect tests/test_bench_gpu.py --deselect tests/test_box_gpu.py --deselect tests/test_c_example_gpu.py --deselect tests/test_config3_gpu.py --deselect tests/test_edge_gpu.py --deselect tests/test_examples_gpu.py > gpurun_out/r4u/gpu_suite_rest.txt 2>&1; rc=$?
echo "rest of suite (from test_fft_gpu on) rc $rc"; tail -6 gpurun_out/r4u/gpu_suite_rest.txt
