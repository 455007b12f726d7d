#!/bin/bash
mkdir -p gpurun_out/r4e
timeout -k 10 1100 python -m pytest tests/test_slab_gpu.py -x -q -k "2048" > gpurun_out/r4e/slab2048.txt 2>&1; rc=$?
echo "slab2048 rc $rc"; tail -15 gpurun_out/r4e/slab2048.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_bench_gpu.py -x -q > gpurun_out/r4e/benchtest.txt 2>&1; rc=$?
echo "bench tests rc $rc"; tail -15 gpurun_out/r4e/benchtest.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4e/bench512.json 2> gpurun_out/r4e/bench512.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4e/bench512.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["regions"])
print("f64", d["f64"]["value"], "config3", d["config3"]["value"], d["config3"]["one_box"]["value"], "sizes", {k:v["value"] for k,v in d["sizes"].items()})
PY
