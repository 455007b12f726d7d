#!/bin/bash
mkdir -p gpurun_out/r4h; OUT=gpurun_out/r4h/rsd_turn_variants.txt; : > $OUT
timeout -k 10 600 python -m pytest tests/test_rsd_turn_gpu.py tests/test_config3_gpu.py -x -q > gpurun_out/r4h/tests.txt 2>&1; rc=$?
echo "tests rc $rc"; tail -6 gpurun_out/r4h/tests.txt
if [ $rc -ge 124 ]; then exit $rc; fi
for rnd in 1 2; do
for v in old p0o6 p1o6 p2o6 p3o6 p0o5 p1o5 p3o5 p2o4; do
  if [ $v = old ]; then lib=bisect/r3head/fastbox_amd/lib/libfastbox_hip.so; else lib=fastbox_amd/lib/variants/lib_$v.so; fi
  [ -f $lib ] || continue
  echo "== $v (round $rnd)" | tee -a $OUT
  FASTBOX_HIP_LIB=$lib timeout -k 10 200 python tools/config3_bench.py 512 1 0 0 2>/dev/null | grep -E "sigma_nl=  0.0|per-kernel" | head -2 | tee -a $OUT
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done; done
