"""Tuning aid (GPU): stand-alone binning of a stored spectrum (k_bin_rows), cubic and cuboid boxes.  [N]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, Wedge
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for scale in (1e3, (1e3, 1.5e3, 2e3)):
    box = CosmoBox(default_cosmo, box_scale=scale, nsamp=N, realise_now=False, precision="f32", rng="device", seed=1)
    dx = box.realise_density()
    dk = box.delta_k
    box.binned_power_spectrum(delta_k=dk)
    eng = box.engine
    eng.sync(); t0 = time.perf_counter()
    for _ in range(10):
        out = box.binned_power_spectrum(delta_k=dk)
    eng.sync(); dt = (time.perf_counter() - t0) / 10
    print("box_scale=%s: binned_power_spectrum(delta_k=...) %.3f ms" % (scale, dt * 1e3))
