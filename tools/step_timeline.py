"""Tuning aid (GPU): how the rate of the benchmarked step develops from a cold start, and where the wall time of a
short run goes (queueing, GPU completion, result handling).
    python tools/step_timeline.py [steps] [streams] [idle_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
idle_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
args = bench.parse(["--steps", str(steps), "--streams", str(nstreams)])
boxes = bench._make_boxes(args, 512, "f32", nstreams, 0, 0)
step = bench._step_fn(boxes, 20)
bench._warm(step, 5)
for rep in range(6):
    torch.cuda.synchronize()
    if idle_ms and rep in (3,):
        time.sleep(idle_ms * 1e-3)          # does an idle gap bring the slow start back?
    t0 = time.perf_counter()
    pend = [step() for _ in range(steps)]
    t1 = time.perf_counter()
    for b in boxes:
        b.engine.sync()
    t2 = time.perf_counter()
    for p in pend:
        p.result()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    print("rep %d steps %d streams %d: queued %.2f ms | GPU done %.2f | all results %.2f | end %.2f  -> %.1f /s"
          % (rep, steps, nstreams, 1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e3 * (t4 - t0), 1e3 * (t5 - t0), steps / (t5 - t0)))
