#!/bin/bash
# usage: plane_batch_sweep.sh N prec reps "B:S B:S ..."
export PYTHONPATH=$PWD
N=$1; P=$2; R=$3
for cfg in $4; do
  B=${cfg%%:*}; S=${cfg##*:}
  if [ "$B" = auto ]; then timeout -k 10 300 python tools/plane_batch_bench.py $N $P $R 2>&1 | grep boxes || exit 1
  else FB_PLANE_BATCH=$B FB_PLANE_STREAMS=$S timeout -k 10 300 python tools/plane_batch_bench.py $N $P $R 2>&1 | grep boxes || exit 1; fi
done
