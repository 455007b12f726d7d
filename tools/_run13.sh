BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
python -m pytest tests/test_sky_gpu.py tests/test_slab_gpu.py -m gpu -x -q > gpurun_out/t13.log 2>&1; tail -5 gpurun_out/t13.log
for NS in 1024 2048; do
python bench.py --no-cpu-baseline --no-extras --nsamp $NS --steps $((NS==1024?30:6)) --warmup 2 --streams 1 --spin-up 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$NS paired s1', round(d['value'],2), d['roofline']['avg_launch_us'])"
python bench.py --no-cpu-baseline --no-extras --nsamp $NS --steps $((NS==1024?30:6)) --warmup 2 --spin-up 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$NS paired s2', round(d['value'],2))"
done
python bench.py --no-cpu-baseline --no-extras --nsamp 1024 --precision f64 --steps 10 --warmup 2 --streams 1 --spin-up 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1024 f64 paired s1', round(d['value'],2))"
python bench.py --no-cpu-baseline --no-extras --steps 200 --streams 1 --all-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512 s1 (contig 256 thr)', round(d['value'],1), d['kernel_ms_per_step'])"
make -C fastbox_amd/csrc clean > /dev/null; make -C fastbox_amd/csrc -j16 CXXFLAGS="$BASE -DFB_NO_XCD_PAIR" > /dev/null 2>&1
for NS in 1024 2048; do
python bench.py --no-cpu-baseline --no-extras --nsamp $NS --steps $((NS==1024?30:6)) --warmup 2 --streams 1 --spin-up 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$NS unpaired s1', round(d['value'],2), d['roofline']['avg_launch_us'])"
done
python bench.py --no-cpu-baseline --no-extras --nsamp 1024 --precision f64 --steps 10 --warmup 2 --streams 1 --spin-up 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1024 f64 unpaired s1', round(d['value'],2))"
for CT in 128 512; do
make -C fastbox_amd/csrc clean > /dev/null; make -C fastbox_amd/csrc -j16 CXXFLAGS="$BASE -DFB_CONTIG_THREADS=$CT" > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-extras --steps 200 --streams 1 --all-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512 s1 contig threads $CT', round(d['value'],1), d['kernel_ms_per_step'])"
done
make -C fastbox_amd/csrc clean > /dev/null; make -C fastbox_amd/csrc -j16 > /dev/null 2>&1
