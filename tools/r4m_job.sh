#!/bin/bash
mkdir -p gpurun_out/r4m
for st in 2 1; do
timeout -k 10 300 python bench.py --nsamp 1024 --no-extras --no-cpu-baseline --steps 20 --warmup 3 --regions 3 --streams $st > gpurun_out/r4m/b1024_s$st.json 2>/dev/null; rc=$?
if [ $rc -ge 124 ]; then exit $rc; fi
python - <<PY
import json
d=json.loads(open("gpurun_out/r4m/b1024_s$st.json").read().strip().splitlines()[-1])
print("1024^3 streams $st: %.1f boxes/s %.3f ms" % (d["value"], d["ms_per_step"]))
for k in ("roofline","roofline_gen","roofline_bin","roofline_z"):
    if k in d: print("  ", k, "%.1f us/launch, launches %s, %.0f GB/s" % (d[k]["avg_launch_us"], d[k].get("launches_timed"), d[k]["achieved"]))
if "kernel_ms_per_step" in d: print("  ", d["kernel_ms_per_step"], d.get("kernel_ms_total_per_step"))
PY
done
timeout -k 10 300 python tools/pass_bench.py 1024 f32 3 2>/dev/null | grep -v amdgpu > gpurun_out/r4m/pass_bench_1024.txt; cat gpurun_out/r4m/pass_bench_1024.txt
