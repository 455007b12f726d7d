"""Tuning aid (GPU): where does the host time of one small-box step go?  python tools/host_profile.py [N]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastbox_amd import CosmoBox, default_cosmo
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
box = CosmoBox(default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, rng="device", seed=1)
def step():
    return box.binned_power_spectrum(delta_x=box.lognormal(box.realise_density()), nbins=20, wait=False)
for _ in range(50): step().result()
pr = cProfile.Profile(); pr.enable()
pend = [step() for _ in range(2000)]
out = [p.result() for p in pend]
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3800])
