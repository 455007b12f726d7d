#!/bin/bash
mkdir -p gpurun_out/r4j; OUT=gpurun_out/r4j/csign.txt; : > $OUT
for N in 512 1024; do for v in default csign; do
  if [ $v = default ]; then unset FASTBOX_HIP_LIB; else export FASTBOX_HIP_LIB=$PWD/fastbox_amd/lib/variants/lib_$v.so; fi
  echo "== pass_bench $N $v" | tee -a $OUT
  timeout -k 10 300 python tools/pass_bench.py $N f32 5 2>/dev/null | grep -E "^y plain|^x plain" | tee -a $OUT
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done; done
for rnd in 1 2 3; do for v in default csign; do
  if [ $v = default ]; then unset FASTBOX_HIP_LIB; else export FASTBOX_HIP_LIB=$PWD/fastbox_amd/lib/variants/lib_$v.so; fi
  for st in 2 1; do
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --streams $st 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512^3 $v streams $st round $rnd: %.1f boxes/s' % d['value'], d['regions']['boxes_per_s'])" | tee -a $OUT
  done
done; done
