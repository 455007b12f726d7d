"""Worker of test_plane_batches_gpu.py: runs every transform chain that uses the plane-batched y/z passes
(fb_fft_launch.inc yz_passes) and prints one SHA-256 per result.  The batching is chosen by the environment
(FB_PLANE_BATCH / FB_PLANE_STREAMS), read once per process, hence a process per setting."""
import hashlib
import sys

import numpy as np

from fastbox_amd import BeamHighpass, CosmoBox, default_cosmo
from oracle import standin

N, precision = int(sys.argv[1]), sys.argv[2]


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(np.asarray(a)).tobytes())
    return h.hexdigest()


box = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision=precision, rng="device", seed=3)
dx = box.realise_density()                                   # generator, then [y, z c2r]
print("density", digest(dx))
print("delta_k", digest(box.delta_k))                        # r2c: [z, y], x
print("pk", digest(*box.binned_power_spectrum(nbins=16)))
dx2 = box.realise_density()
ln = box.lognormal(dx2)
print("pk_lognormal", digest(*box.binned_power_spectrum(delta_x=ln, nbins=16)))   # [y, z c2r2c + exp, y], x + bins
print("density2", digest(dx2))
print("lognormal", digest(ln))
vel = box.realise_velocity()
print("velocity_z", digest(box.to_real(vel[2])))
lazy = box.apply_transfer_fn(box.to_k(dx2), BeamHighpass(kpar0=0.01, kperp0=0.1, power=2.))
print("pk_filtered", digest(*box.binned_power_spectrum(delta_x=lazy.real, nbins=16)))   # [z, y], x * T + bins
print("filtered", digest(lazy.real))                         # c2r of the stored filtered spectrum
print("callable_filter", digest(box.apply_transfer_fn(box.delta_k, standin.beam_highpass)))   # full complex c2c
np.random.seed(4)
box_np = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision=precision)
print("numpy_noise_density", digest(box_np.realise_density()))    # colour + c2r: x, [y, z]
