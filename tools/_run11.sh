python -m pytest tests -m gpu -x -q > gpurun_out/t11.log 2>&1; tail -12 gpurun_out/t11.log
for ST in 0 64 128 192 256; do echo "stagger $ST"; python tools/pass_bench.py 512 f32 $ST 2>/dev/null | grep -E "x gen  |x bin  |y plain  "; done
