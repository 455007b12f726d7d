#!/bin/bash
# boxes per GPU (streams) x planes per batch x plane streams, three repeats each (512^3, 200 steps)
#   bash tools/streams_sweep.sh > gpurun_out/streams_sweep.txt
for S in 2 3; do
for PB in 48 64 96; do
for PS in 1 2; do
  echo -n "streams=$S plane_batch=$PB plane_streams=$PS:"
  for rep in 1 2 3; do
    python bench.py --no-cpu-baseline --no-extras --steps 200 --streams $S --plane-batch $PB --plane-streams $PS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(' %.1f' % d['value'], end='')"
  done
  echo
done; done; done
