#!/bin/bash
# round 4: CU lanes (fb_set_cu_split) -- headline step at 512^3, boxes in flight x CUs per XCD of the memory lane x plane batch
mkdir -p gpurun_out/r4w; OUT=gpurun_out/r4w/cu_split_sweep2.txt; : > $OUT
for cfg in "0 2 64" "16 2 128" "16 3 128" "18 3 128" "20 3 128" "18 2 128" "16 3 64" "22 3 128" "18 4 128"; do
  set -- $cfg; m=$1; st=$2; pb=$3
  line=$(FB_CU_SPLIT=$m timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 --regions 3 --streams $st --plane-batch $pb 2>gpurun_out/r4w/err_${m}_${st}.txt | tail -1); rc=$?
  if [ $rc -ge 124 ]; then echo "timeout at $cfg" | tee -a $OUT; exit $rc; fi
  python - "$line" <<PY | tee -a $OUT
import json,sys
try:
    d=json.loads(sys.argv[1])
    print("memory lane $m CUs/XCD, $st boxes in flight, plane batch $pb: %7.1f boxes/s %s" % (d["value"], d["regions"]["boxes_per_s"]))
except Exception as e:
    print("memory lane $m, $st boxes: FAILED", e, sys.argv[1][:300])
PY
done
