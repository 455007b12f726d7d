#!/bin/bash
# GPU box: tools/phase_timeline.py for the generator pass, a y pass and the binning pass, with the tile rotation of
# k_fft_strided off and on (diagnostic -DFB_STAMPS builds; leaves the DEFAULT build in place).
#   bash tools/run_phase_timeline.sh > gpurun_out/phase_timeline.txt
set -e
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
for V in "-DFB_NO_TILE_ROTATE" ""; do
    make -C fastbox_amd/csrc clean > /dev/null
    make -C fastbox_amd/csrc -j16 CXXFLAGS="$BASE -DFB_STAMPS $V" > /dev/null 2>&1
    for AM in "0 1" "1 0" "0 2"; do
        echo "==== build [-DFB_STAMPS $V]  phase_timeline.py $AM"
        python tools/phase_timeline.py $AM 2>&1 | grep -v amdgpu.ids
    done
done
make -C fastbox_amd/csrc clean > /dev/null
make -C fastbox_amd/csrc -j16 > /dev/null 2>&1
