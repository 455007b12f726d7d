#!/bin/bash
# GPU box: the evidence behind bench.py's roofline object, written to gpurun_out/prof_<tag>/.
#   bash tools/collect_profiles.sh r02
# 1. rocprofv3 --kernel-trace --stats of `bench.py --streams 1` (one box: per-kernel average durations of kernels
#    running alone -- what bench.py's own un-overlapped roofline pass measures with HIP events) and of the default
#    command (two boxes on two streams: durations of kernels that share the chip, for the record)
# 2. two separate --pmc passes over the default command (FETCH_SIZE, WRITE_SIZE: KiB per dispatch; FETCH_SIZE is doubled on
#    gfx950 when it is compared with bytes, see MI355X_MICROARCH.md) summarised per kernel
# Afterwards, in the build container: copy the summaries to profiles/<tag>_* and write profiles/current.json with the
# commit they were collected at (tools/stamp_profiles.py).
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 $R/bench.py $ARGS --streams 1 > $OUT/bench_streams1_under_rocprof.json 2> $OUT/trace1.err
cp $(ls $OUT/trace1/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_streams1.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- python3 $R/bench.py $ARGS --no-one-box-pass > $OUT/bench_under_rocprof.json 2> $OUT/trace2.err
cp $(ls $OUT/trace2/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
# the launches of the default command's roofline pass, every kernel alone: one box, the 64-plane batches of the two-box run
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace3 -- python3 $R/bench.py $ARGS --streams 1 --plane-batch 64 > $OUT/bench_streams1_pb64_under_rocprof.json 2> $OUT/trace3.err
cp $(ls $OUT/trace3/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_streams1_pb64.csv
# BASELINE config 3 (one box): the fused redshift-space z pass k_rsd_turn and the rest of that chain, every kernel alone
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace4 -- python3 $R/tools/config3_bench.py 512 > $OUT/config3_under_rocprof.txt 2> $OUT/trace4.err
cp $(ls $OUT/trace4/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_config3.csv
for C in FETCH_SIZE WRITE_SIZE; do
    # the default command (two boxes, 64-plane batches at 512^3): the counters are per dispatch, and the profiler runs
    # the dispatches of a counter pass one at a time
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $ARGS --no-one-box-pass --spin-up 0 --steps 10 --warmup 2 > /dev/null 2> $OUT/pmc_$C.err
done
python3 $R/tools/pmc_summary.py $OUT/pmc_fetch_write_summary.json $(ls $OUT/pmc_*/*/*counter_collection.csv) > $OUT/pmc_summary.txt
rm -rf $OUT/trace1 $OUT/trace2 $OUT/trace3 $OUT/trace4 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench.err
python3 $R/bench.py $ARGS --streams 1 > $OUT/bench_streams1.json 2>> $OUT/bench.err
echo done
