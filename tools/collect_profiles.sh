#!/bin/bash
# GPU box: the evidence behind bench.py's roofline object, written to gpurun_out/prof_<tag>/.
#   bash tools/collect_profiles.sh r01e
# 1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel average durations)
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: KiB per dispatch; FETCH_SIZE is doubled on
#    gfx950 when it is compared with bytes, see MI355X_MICROARCH.md) summarised per kernel
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2> $OUT/pmc_$C.err
done
python3 $R/tools/pmc_summary.py $OUT/pmc_fetch_write_summary.json $(ls $OUT/pmc_*/*/*counter_collection.csv) > $OUT/pmc_summary.txt
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo done
