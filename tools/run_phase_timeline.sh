set -e
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
make -C fastbox_amd/csrc clean > /dev/null
make -C fastbox_amd/csrc -j16 CXXFLAGS="$BASE -DFB_STAMPS" > /dev/null 2>&1
FB_STAMPS_DUMP=gpurun_out/stamps_gen.npy python tools/phase_timeline.py 0 1 > gpurun_out/phase_gen.txt 2>&1 || true
FB_STAMPS_DUMP=gpurun_out/stamps_y.npy python tools/phase_timeline.py 1 0 > gpurun_out/phase_y.txt 2>&1 || true
FB_STAMPS_DUMP=gpurun_out/stamps_bin.npy python tools/phase_timeline.py 0 2 > gpurun_out/phase_bin.txt 2>&1 || true
make -C fastbox_amd/csrc clean > /dev/null
make -C fastbox_amd/csrc -j16 > /dev/null 2>&1
