#!/bin/bash
mkdir -p gpurun_out/r4o
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/r4o/gpu_suite.txt 2>&1; rc=$?
echo "suite rc $rc"; tail -8 gpurun_out/r4o/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
