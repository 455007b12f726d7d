#!/bin/bash
# VERDICT r3 Next 4: where did 256^3 lose 6-8 % in round 3?  bench.py of each exported commit (bisect/<commit>/, its own
# library) and of the head, interleaved, on one GPU box.   bash tools/bisect256.sh <outfile> [rounds]
OUT=${1:-gpurun_out/bisect256.txt}; ROUNDS=${2:-2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
: > $OUT
for r in $(seq 1 $ROUNDS); do
  for c in 0e226fa 04aa0fb 043b8de 2079f1a 102fb74 HEAD; do
    for s in 2 1; do
      if [ $c = HEAD ]; then d=$ROOT; else d=$ROOT/bisect/$c; fi
      line=$(cd $d && timeout -k 10 300 python bench.py --nsamp 256 --no-extras --no-cpu-baseline --steps 400 --streams $s 2>/dev/null | tail -1)
      v=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.1f boxes/s  %.4f ms' % (d['value'], d['ms_per_step']))" "$line" 2>/dev/null)
      echo "round $r  $c  streams $s  $v" | tee -a $OUT
    done
  done
done
