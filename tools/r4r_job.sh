#!/bin/bash
mkdir -p gpurun_out/r4r; OUT=gpurun_out/r4r/sweep2048b.txt; : > $OUT
for rnd in 1 2; do for cfg in "0 4" "1 4" "1 8" "1 16" "1 32" "0 16"; do
  set -- $cfg; mask=$1; pb=$2
  line=$(FB_WIDE_ROWS=$mask timeout -k 10 300 python bench.py --nsamp 2048 --no-extras --no-cpu-baseline --steps 12 --warmup 2 --regions 3 --plane-batch $pb 2>/dev/null | tail -1); rc=$?
  if [ $rc -ge 124 ]; then echo timeout; exit $rc; fi
  python - "$line" <<PY | tee -a $OUT
import json,sys
d=json.loads(sys.argv[1])
print("round $rnd  wide mask $mask  plane batch %2d: %6.2f boxes/s %s" % ($pb, d["value"], d["regions"]["boxes_per_s"]))
PY
done; done
