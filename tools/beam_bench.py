"""Beam convolution of an N^3 cube on the device (BeamModel.convolve_fft: zero-padded 2N x 2N transforms per channel;
convolve_real: N x N circular), time per call and per kernel class.
    python tools/beam_bench.py [N]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import BeamModel, CosmoBox, default_cosmo

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
box = CosmoBox(cosmo=default_cosmo, box_scale=(2e3, 2e3, 1e3), nsamp=N, redshift=0.6, realise_now=False,
               precision="f32", rng="device", seed=2)
eng = box.engine
dx = box.realise_density()
ang_x, ang_y = box.pixel_array()
x, y = np.meshgrid(ang_x, ang_y, indexing="ij")
sig = 0.1 * (ang_x[-1] - ang_x[0])
cube = eng.upload(np.repeat(np.exp(-0.5 * (x ** 2 + y ** 2) / sig ** 2)[:, :, None], N, axis=2), "real")


class GaussBeam(BeamModel):
    def beam_cube(self, pol=None):
        return cube


beam = GaussBeam(box)
for name, fn in (("convolve_fft (2N x 2N zero-padded)", beam.convolve_fft), ("convolve_real (N x N circular)", beam.convolve_real)):
    out = fn(dx); eng.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        out = fn(dx)
    eng.sync()
    dt = (time.perf_counter() - t0) / 5
    print("N=%d %s: %.2f ms per cube, std in / out %.4f / %.4f" % (N, name, 1e3 * dt, float(np.std(np.asarray(dx)[::8, ::8, ::8])),
                                                                  float(np.std(np.asarray(out)[::8, ::8, ::8]))))
