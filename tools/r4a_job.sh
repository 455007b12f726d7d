#!/bin/bash
mkdir -p gpurun_out/r4a
timeout -k 10 240 ./tools/plane_team.bin > gpurun_out/r4a/plane_team.txt 2>&1; rc=$?
echo "plane_team rc $rc"; tail -40 gpurun_out/r4a/plane_team.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 python tools/lognormal_dev.py > gpurun_out/r4a/lognormal_dev.txt 2>&1; rc=$?
echo "lognormal_dev rc $rc"; cat gpurun_out/r4a/lognormal_dev.txt
if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/bisect256.sh gpurun_out/r4a/bisect256.txt 2
