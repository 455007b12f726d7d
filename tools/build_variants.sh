#!/bin/bash
# Tuning aid (CPU container: hipcc cross-compiles): build the library several times with different -D flags into
# fastbox_amd/lib/variants/lib_<name>.so, which travel to the GPU box with the snapshot (no compile time there).
#   bash tools/build_variants.sh name1 "-DFLAG_A -DFLAG_B" name2 "-DFLAG_C" ...
# Run one with  FASTBOX_HIP_LIB=fastbox_amd/lib/variants/lib_<name>.so python tools/pass_bench.py 512
set -e
cd "$(dirname "$0")/.."
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
mkdir -p fastbox_amd/lib/variants
while [ $# -ge 2 ]; do
    NAME="$1"; FLAGS="$2"; shift 2
    OBJ="/tmp/fb_variant_obj_$NAME"
    mkdir -p "$OBJ"
    make -C fastbox_amd/csrc -j"${JOBS:-8}" CXXFLAGS="$BASE $FLAGS" OBJDIR="$OBJ" OUT="../lib/variants/lib_$NAME.so" > "$OBJ/build.log" 2>&1 \
        || { tail -30 "$OBJ/build.log"; exit 1; }
    echo "built lib_$NAME.so [$FLAGS]"
done
