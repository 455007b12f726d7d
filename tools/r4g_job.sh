#!/bin/bash
mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests/test_slab_gpu.py tests/test_box_gpu.py tests/test_bench_gpu.py -x -q -k "communicator or queued or pending or two_ranks or power" > gpurun_out/r4g/tests.txt 2>&1; rc=$?
echo "tests rc $rc"; tail -8 gpurun_out/r4g/tests.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/host_profile.py 256 3000 2 > gpurun_out/r4g/host_profile_256.txt 2>&1; echo "host profile rc $?"; head -16 gpurun_out/r4g/host_profile_256.txt
for s in 2 1; do timeout -k 10 300 python bench.py --nsamp 256 --no-extras --no-cpu-baseline --steps 400 --streams $s 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256^3 streams $s', d['value'], d['regions']['boxes_per_s'])"; done
