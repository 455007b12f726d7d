#!/bin/bash
# streams x plane-batch sweep (512^3, 200 steps)
for S in 2 3; do
for PS in 1 2; do
for B in 32 48 64 96 128; do
  echo -n "streams=$S plane_streams=$PS batch=$B: "
  FB_PLANE_STREAMS=$PS FB_PLANE_BATCH=$B python bench.py --no-cpu-baseline --steps 200 --streams $S 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"
done; done; done
