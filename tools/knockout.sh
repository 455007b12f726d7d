#!/bin/bash
# Tuning aid (GPU box): rebuild the library with one part of the fused generator pass knocked out at a
# time and time the pass (tools/pass_bench.py).  Leaves the DEFAULT build in place when it finishes.
#   bash tools/knockout.sh > gpurun_out/knockout.txt
# Variants: KNOCKOUT_VARIANTS="... -DFB_EXPERIMENT_NOFFT ..." ('@' joins the flags of one variant); -DFB_EXPERIMENT_NOFFT
# drops the pass's transform (profiles/r03_gen_knockout_1024_2048.txt was made with tools/build_variants.sh instead).
set -e
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
for V in ${KNOCKOUT_VARIANTS:-"" "-DFB_EXPERIMENT_NOAMP" "-DFB_EXPERIMENT_NOBM" "-DFB_PHILOX_ROUNDS=1" \
         "-DFB_EXPERIMENT_NOAMP@-DFB_EXPERIMENT_NOBM@-DFB_PHILOX_ROUNDS=1"}; do
    V="${V//@/ }"
    make -C fastbox_amd/csrc clean > /dev/null
    make -C fastbox_amd/csrc -j16 CXXFLAGS="$BASE $V" > /dev/null 2>&1
    echo "== variant: [$V]"
    if [ -n "$KNOCKOUT_CONFIG3" ]; then
        python tools/config3_bench.py 512 | grep -v amdgpu.ids | sed -n 1,2p
    elif [ -n "$KNOCKOUT_BENCH" ]; then
        python bench.py --all-kernel-events --no-cpu-baseline --steps 40 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['kernel_ms_per_step'])"
    else
        python tools/pass_bench.py 512 | grep -E "gen|bin|plain, no"
    fi
done
make -C fastbox_amd/csrc clean > /dev/null
make -C fastbox_amd/csrc -j16 > /dev/null 2>&1
