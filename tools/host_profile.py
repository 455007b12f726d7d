"""Where does the HOST spend its time per benchmarked step?  cProfile over bench.py's step (realise_density ->
lognormal -> binned_power_spectrum(wait=False), results fetched at the end) at a size where the GPU is not the limit.
    python tools/host_profile.py [N=256] [steps=3000] [boxes=2]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
nbox = int(sys.argv[3]) if len(sys.argv) > 3 else 2
args = bench.parse(["--nsamp", str(N), "--streams", str(nbox)])
boxes = bench._make_boxes(args, N, "f32", nbox, 0, 0)
step = bench._step_fn(boxes, 20)
bench._warm(step, 50)
t0 = time.perf_counter()
pend = [step() for _ in range(steps)]
t_issue = time.perf_counter() - t0
out = [p.result() for p in pend]
t_all = time.perf_counter() - t0
print("N = %d, %d boxes: %.1f us per step to issue, %.1f us per step in all (%.0f boxes/s)" %
      (N, nbox, 1e6 * t_issue / steps, 1e6 * t_all / steps, steps / t_all))
pr = cProfile.Profile()
pr.enable()
pend = [step() for _ in range(steps)]
out = [p.result() for p in pend]
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
