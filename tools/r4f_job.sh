#!/bin/bash
mkdir -p gpurun_out/r4f
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4f/gpu_suite.txt 2>&1; rc=$?
echo "suite rc $rc"; tail -25 gpurun_out/r4f/gpu_suite.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/host_profile.py 256 3000 2 > gpurun_out/r4f/host_profile_256.txt 2>&1; echo "host profile rc $?"; head -40 gpurun_out/r4f/host_profile_256.txt
