#!/bin/bash
mkdir -p gpurun_out/r4l
timeout -k 10 300 python bench.py --steps 2000 --warmup 5 --regions 5 --no-extras --no-cpu-baseline > gpurun_out/r4l/soak.json 2>/dev/null; rc=$?; echo "soak rc $rc"
if [ $rc -ge 124 ]; then exit $rc; fi
python -c "
import json; d=json.loads(open('gpurun_out/r4l/soak.json').read().strip().splitlines()[-1])
print('%.1f boxes/s (median of %d regions of %d steps: %s), finite %s, repeats %d, pipeline frac %.3f / %.3f' % (d['value'], d['regions']['count'], d['steps'], d['regions']['boxes_per_s'], d['finite'], d['lognormal_repeats'], d['pipeline_roofline']['frac'], d['pipeline_roofline']['frac_moved']))"
timeout -k 10 600 python tools/montecarlo_cov.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4l/montecarlo.txt; echo "mc rc $?"; head -8 gpurun_out/r4l/montecarlo.txt
