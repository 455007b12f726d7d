#!/bin/bash
mkdir -p gpurun_out/r4i; OUT=gpurun_out/r4i/z2048.txt; : > $OUT
for rnd in 1 2; do for v in tw2off default; do
  if [ $v = default ]; then unset FASTBOX_HIP_LIB; else export FASTBOX_HIP_LIB=$PWD/fastbox_amd/lib/variants/lib_$v.so; fi
  line=$(timeout -k 10 300 python bench.py --nsamp 2048 --no-extras --no-cpu-baseline --steps 8 --warmup 2 --regions 3 2>/dev/null | tail -1); rc=$?
  if [ $rc -ge 124 ]; then echo timeout; exit $rc; fi
  python - "$line" <<PY | tee -a $OUT
import json,sys
d=json.loads(sys.argv[1])
print("$v round $rnd: %6.2f boxes/s  %6.2f ms | y %6.1f us/launch  gen %7.1f  bin %7.1f  z %6.1f (%.0f GB/s)" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline_gen"]["avg_launch_us"], d["roofline_bin"]["avg_launch_us"], d["roofline_z"]["avg_launch_us"], d["roofline_z"]["achieved"]))
PY
done; done
unset FASTBOX_HIP_LIB
timeout -k 10 1000 python -m pytest tests/test_lognormal_gpu.py tests/test_edge_gpu.py tests/test_slab_gpu.py tests/test_pass_schedule_gpu.py -x -q -k "2048" > gpurun_out/r4i/tests2048.txt 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r4i/tests2048.txt
