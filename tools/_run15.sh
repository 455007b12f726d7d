BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
python -m pytest tests/test_fft_gpu.py tests/test_box_gpu.py tests/test_edge_gpu.py tests/test_plane_batches_gpu.py -m gpu -x -q > gpurun_out/t15.log 2>&1; tail -6 gpurun_out/t15.log
for rep in 1 2; do
python bench.py --no-cpu-baseline --no-extras --steps 200 --streams 1 --all-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new z', round(d['value'],1), d['kernel_ms_per_step'])"
done
python bench.py --no-cpu-baseline --no-extras --steps 200 --streams 1 --all-kernel-events --gaussian-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new z, gaussian only', round(d['value'],1), d['kernel_ms_per_step'])"
for rep in 1 2; do python bench.py --no-cpu-baseline --no-extras --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('s2 200', round(d['value'],1))"; done
make -C fastbox_amd/csrc clean > /dev/null; make -C fastbox_amd/csrc -j16 CXXFLAGS="$BASE -DFB_LIBM_EXP" > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-extras --steps 200 --streams 1 --all-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('libm exp', round(d['value'],1), d['kernel_ms_per_step'])"
make -C fastbox_amd/csrc clean > /dev/null; make -C fastbox_amd/csrc -j16 > /dev/null 2>&1
