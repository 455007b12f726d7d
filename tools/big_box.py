"""GPU sanity run at large N: realisation + P(k) (fused), Parseval, fused vs stand-alone binning."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
t0 = time.time()
box = CosmoBox(cosmo=default_cosmo, box_scale=2e3, nsamp=N, realise_now=False, precision=prec, rng="device", seed=1)
dx = box.realise_density()
fused = box.binned_power_spectrum(delta_x=dx, nbins=20)
box.engine.sync(); t1 = time.time()
for _ in range(3):
    dx = box.realise_density()
    fused = box.binned_power_spectrum(delta_x=box.lognormal(dx), nbins=20)
box.engine.sync(); t2 = time.time()
dx = box.realise_density()
fused = box.binned_power_spectrum(delta_x=dx, nbins=20)
s1, s2 = box.test_parseval()
plain = box.binned_power_spectrum(nbins=20)
m = ~np.isnan(plain[1])
print("N", N, prec, "first box %.2f s, then %.3f s per gen+lognormal+P(k)" % (t1 - t0, (t2 - t1) / 3))
print("parseval ratio", s1 / s2, " fused/plain max rel diff", np.max(np.abs(fused[1][m] / plain[1][m] - 1)))
assert abs(s1 / s2 - 1) < 1e-4 and np.max(np.abs(fused[1][m] / plain[1][m] - 1)) < 1e-5
print("OK")
