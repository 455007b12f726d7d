"""Tuning aid (GPU): time single strided FFT passes with HIP events.  python tools/pass_bench.py [N]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, _lib
from fastbox_amd.device import HALF

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
box = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision=prec, rng="device")
eng = box.engine
dx = box.realise_density()
box.binned_power_spectrum(delta_x=dx)            # sets bins/thresholds
h = eng.empty(HALF)
nbytes = 2.0 * N * N * (N // 2 + 1) * (8 if prec == "f32" else 16)
stag = int(sys.argv[3]) if len(sys.argv) > 3 else 0
_lib.call("fb_set_tuning", eng._plan, stag, stag, stag)
print("stagger", stag)
for name, axis, mode, traffic in (("y plain", 1, 0, nbytes), ("x plain", 0, 0, nbytes), ("x gen  ", 0, 1, nbytes / 2),
                                  ("x bin  ", 0, 2, nbytes / 2), ("y plain, no memory traffic", 1, 10, 0.0),
                                  ("x gen,   no memory traffic", 0, 11, 0.0), ("x bin,   no memory traffic", 0, 12, 0.0)):
    for rep in range(2):
        eng.profile_start()
        for _ in range(10):
            _lib.call("fb_debug_strided_pass", eng._plan, h.ptr, axis, mode, eng.stream)
        prof = eng.profile_stop()
    ms = sum(v[0] for v in prof.values()) / 10
    print("%s  %8.1f us   %7.0f GB/s (algorithmic)" % (name, ms * 1e3, traffic / ms / 1e6))
