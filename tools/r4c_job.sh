#!/bin/bash
mkdir -p gpurun_out/r4c
timeout -k 10 900 python -m pytest tests/test_pass_schedule_gpu.py -x -q -k "2048" > gpurun_out/r4c/test2048.txt 2>&1; rc=$?
echo "test rc $rc"; tail -15 gpurun_out/r4c/test2048.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python tools/pass_bench.py 2048 f32 3 > gpurun_out/r4c/pass_bench_2048.txt 2>&1; rc=$?
echo "pass_bench rc $rc"; cat gpurun_out/r4c/pass_bench_2048.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --nsamp 2048 --no-extras --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r4c/bench2048.txt 2>&1; echo "bench rc $?"; tail -2 gpurun_out/r4c/bench2048.txt
timeout -k 10 200 ./tools/plane_team.bin > gpurun_out/r4c/plane_team5.txt 2>&1; echo rc $?; grep -E "^delay 0|time-outs" gpurun_out/r4c/plane_team5.txt
