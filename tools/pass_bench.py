"""Tuning aid (GPU): time single strided FFT passes with HIP events, both schedules of each pass class interleaved in
one process (cdna_hip_programming.md rule 24).  python tools/pass_bench.py [N] [f32|f64] [rounds]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, _lib
from fastbox_amd.device import HALF

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
box = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision=prec, rng="device")
eng = box.engine
dx = box.realise_density()
box.binned_power_spectrum(delta_x=dx)            # sets bins/thresholds
h = eng.empty(HALF)
nbytes = 2.0 * N * N * (N // 2) * (8 if prec == "f32" else 16)
reps = 10 if N <= 512 else 3
cases = (("y plain", 1, 0, nbytes), ("x plain", 0, 0, nbytes), ("x gen  ", 0, 1, nbytes / 2), ("x bin  ", 0, 2, nbytes / 2),
         ("y plain, no memory traffic", 1, 10, 0.0), ("x gen,   no memory traffic", 0, 11, 0.0),
         ("x bin,   no memory traffic", 0, 12, 0.0))
staggers = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]      # resident form: start-up delay of the
res = {}                                                                                 # second workgroup of a CU, x64 cycles
variants = [(0, 0)] + [(1, st) for st in staggers]
if N == 2048 and prec == "f32":      # the second number is then the tile's row segment in bytes (fb_set_tile_rows)
    variants = [(0, 64), (1, 64), (0, 128), (1, 128)]
for rnd in range(rounds + 1):
    for name, axis, mode, traffic in cases:
        for sched, stag in variants:
            eng.set_pass_schedule(sched, sched, sched)
            if N == 2048 and prec == "f32":
                eng.set_tile_rows(stag)
            eng.profile_start()
            for _ in range(reps):
                _lib.call("fb_debug_strided_pass", eng._plan, h.ptr, axis, mode, eng.stream)
            prof = eng.profile_stop()
            if rnd:                                  # round 0: warm-up
                res.setdefault((name, sched, stag), []).append(sum(v[0] for v in prof.values()) / reps)
print("N = %d %s; per pass: median (min) over %d rounds of %d launches; schedule 0 = one workgroup per tile, 1 = resident" % (N, prec, rounds, reps))
for name, axis, mode, traffic in cases:
    row = []
    for sched, stag in variants:
        ms = np.array(res[(name, sched, stag)])
        row.append("s%d/%d %7.1f (%7.1f)" % (sched, stag, np.median(ms) * 1e3, ms.min() * 1e3))
    print("%-28s %s" % (name, "  ".join(row)))
