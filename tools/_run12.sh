BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
python -m pytest tests/test_sky_gpu.py tests/test_slab_gpu.py -m gpu -x -q > gpurun_out/t12.log 2>&1; tail -5 gpurun_out/t12.log
for rep in 1 2; do python bench.py --no-cpu-baseline --no-extras --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default s2 200', round(d['value'],1))"; done
for V in "-DFB_GEN_STORE_AUX=2" "-DFB_BIN_LOAD_AUX=2" "-DFB_GEN_STORE_AUX=2 -DFB_BIN_LOAD_AUX=2"; do
  make -C fastbox_amd/csrc clean > /dev/null; make -C fastbox_amd/csrc -j16 CXXFLAGS="$BASE $V" > /dev/null 2>&1
  for rep in 1 2; do python bench.py --no-cpu-baseline --no-extras --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V s2 200', round(d['value'],1))"; done
done
make -C fastbox_amd/csrc clean > /dev/null; make -C fastbox_amd/csrc -j16 > /dev/null 2>&1
