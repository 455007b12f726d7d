"""Headline step (gen + log-normal + P(k)) at a given size / precision under the plane-batch setting of the
environment (FB_PLANE_BATCH, FB_PLANE_STREAMS: fb_fft_launch.inc yz_passes); prints boxes/s and one P(k) value
(identical across settings).  Driven by tools/plane_batch_sweep.sh.
    python tools/plane_batch_bench.py N f32|f64 repetitions"""
import sys, time, numpy as np
import torch
from fastbox_amd import CosmoBox, default_cosmo
N = int(sys.argv[1]); prec = sys.argv[2]; reps = int(sys.argv[3])
box = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision=prec, rng="device", seed=7)
def step():
    dx = box.realise_density()
    ln = box.lognormal(dx)
    return box.binned_power_spectrum(delta_x=ln, nbins=20)
kc, pk, err = step()
pk = np.asarray(pk)
for _ in range(3): r = step()
np.asarray(r[1]); torch.cuda.synchronize()
t = time.time()
for _ in range(reps): r = step()
np.asarray(r[1]); torch.cuda.synchronize()
import os
print(N, prec, "B=%s S=%s" % (os.environ.get("FB_PLANE_BATCH", "auto"), os.environ.get("FB_PLANE_STREAMS", "1")),
      "%.2f boxes/s" % (reps / (time.time() - t)), repr(float(pk[3])), flush=True)
