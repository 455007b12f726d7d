#!/bin/bash
# 2048^3 headline step under the wide-row mask (FB_WIDE_ROWS: 1 plain, 2 generator, 4 binning) x plane batch x boxes per GPU
mkdir -p gpurun_out/r4d; OUT=gpurun_out/r4d/sweep2048.txt; : > $OUT
timeout -k 10 900 python -m pytest tests/test_pass_schedule_gpu.py -x -q -k "2048" > gpurun_out/r4d/test2048.txt 2>&1; rc=$?
echo "test rc $rc"; tail -3 gpurun_out/r4d/test2048.txt
if [ $rc -ge 124 ]; then exit $rc; fi
for streams in 2 1; do for mask in 0 1 7; do for pb in 0 4 8 16; do
  if [ $mask = 7 ] && [ $pb != 4 ]; then continue; fi
  line=$(FB_WIDE_ROWS=$mask timeout -k 10 300 python bench.py --nsamp 2048 --no-extras --no-cpu-baseline --steps 8 --warmup 2 --streams $streams --plane-batch $pb 2>/dev/null | tail -1); rc=$?
  if [ $rc -ge 124 ]; then echo "timeout"; exit $rc; fi
  python - "$line" <<PY | tee -a $OUT
import json,sys
d=json.loads(sys.argv[1])
print("boxes/GPU $streams  wide mask $mask  plane batch %2d: %6.2f boxes/s  %6.2f ms  | y %6.1f us/launch  gen %7.1f  bin %7.1f  z %6.1f" % ($pb, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline_gen"]["avg_launch_us"], d["roofline_bin"]["avg_launch_us"], d["roofline_z"]["avg_launch_us"]))
PY
done; done; done
