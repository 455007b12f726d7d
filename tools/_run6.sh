python -m pytest tests -m gpu -x -q > gpurun_out/t6.log 2>&1; tail -25 gpurun_out/t6.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b6_driver.json 2> gpurun_out/b6.err; python -c "
import json; d=json.load(open('gpurun_out/b6_driver.json')); print('driver cmd', round(d['value'],1), d['ms_per_step'], d['roofline']['frac'], d['config3'].get('ms_per_chain'), d['f64'].get('value'))"
for PB in 48 64 96; do for PS in 1 2; do
python bench.py --no-cpu-baseline --no-extras --steps 200 --plane-batch $PB --plane-streams $PS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('s2 pb$PB ps$PS', round(d['value'],1))"
done; done
python bench.py --no-cpu-baseline --no-extras --steps 200 --streams 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('s3 auto', round(d['value'],1))"
python bench.py --no-cpu-baseline --no-extras --steps 200 --streams 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('s1', round(d['value'],1))"
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('s2 20 steps no extras', round(d['value'],1))"
