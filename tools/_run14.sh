python -m pytest tests/test_sky_gpu.py tests/test_slab_gpu.py tests/test_bench_gpu.py -m gpu -x -q > gpurun_out/t14.log 2>&1; tail -5 gpurun_out/t14.log
for rep in 1 2; do
python bench.py --no-cpu-baseline --no-extras --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('s2 200', round(d['value'],1))"
python bench.py --no-cpu-baseline --no-extras --steps 200 --stream-priorities 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('s2 200 priorities', round(d['value'],1))"
python bench.py --no-cpu-baseline --no-extras --steps 200 --streams 3 --stream-priorities 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('s3 200 priorities', round(d['value'],1))"
done
