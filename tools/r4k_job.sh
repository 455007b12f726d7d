#!/bin/bash
mkdir -p gpurun_out/r4k; OUT=gpurun_out/r4k/r32.txt; : > $OUT
timeout -k 10 900 python -m pytest tests/test_box_gpu.py tests/test_pass_schedule_gpu.py tests/test_plane_batches_gpu.py -x -q -k "512 or headline or benchmarked or plane" > gpurun_out/r4k/tests.txt 2>&1; rc=$?
echo "tests rc $rc"; tail -12 gpurun_out/r4k/tests.txt
if [ $rc -ge 124 ]; then exit $rc; fi
for v in r32off default; do
  if [ $v = default ]; then unset FASTBOX_HIP_LIB; else export FASTBOX_HIP_LIB=$PWD/fastbox_amd/lib/variants/lib_$v.so; fi
  echo "== pass_bench 512 $v" | tee -a $OUT
  timeout -k 10 300 python tools/pass_bench.py 512 f32 5 2>/dev/null | grep -E "^x gen|^x bin" | tee -a $OUT
done
for rnd in 1 2; do for v in r32off r32gen r32bin default; do
  if [ $v = default ]; then unset FASTBOX_HIP_LIB; else export FASTBOX_HIP_LIB=$PWD/fastbox_amd/lib/variants/lib_$v.so; fi
  for st in 2 1; do
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --streams $st 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512^3 $v streams $st round $rnd: %.1f boxes/s' % d['value'], d.get('roofline_gen',{}).get('avg_launch_us'), d.get('roofline_bin',{}).get('avg_launch_us'))" | tee -a $OUT
  done
done; done
