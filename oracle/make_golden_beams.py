"""
TEST INFRASTRUCTURE ONLY.  Golden vectors for the beam-convolution step, captured from the reference's own
fastbox/beams.py (loaded by path, stand-in pyccl) on top of the reference CosmoBox: BeamModel.convolve_fft and
convolve_real with the base class's uniform beam and with a subclass whose beam_cube returns the fixture's own
Gaussian beam (an input, stored in the fixture).
Run in the build container only:   python -m oracle.make_golden_beams
Fixtures hold seeds, inputs and expected outputs, never reference source.
"""
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import beam_oracle, standin                                  # noqa: E402
from oracle.ref_loader import load_reference_box, load_reference_module   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CASES = [("beam_n16", 16, (3e3, 3e3, 3e3), 0.8, 21, True), ("beam_n32", 32, (4e3, 4e3, 2e3), 0.5, 22, False)]


def main():
    ref = load_reference_box()
    beams = load_reference_module("fastbox/beams.py", "_fastbox_reference_beams")
    for name, N, scale, z, seed, with_real in CASES:
        np.random.seed(seed)
        box = ref.CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=scale, nsamp=N, redshift=z, realise_now=False)
        ang_x, ang_y = box.pixel_array()
        freqs = box.freq_array()
        cube = beam_oracle.test_beam_cube(ang_x, ang_y, freqs, fwhm_deg=0.35 * (ang_x[-1] - ang_x[0]))

        class FixtureBeam(beams.BeamModel):
            def beam_cube(self, pol=None):
                return cube

        field = 0.2 + np.random.normal(size=(N, N, N)) * np.linspace(0.5, 1.5, N)[np.newaxis, np.newaxis, :]
        out = dict(numpy_version=np.__version__, scipy_version=scipy.__version__, N=N,
                   box_scale=np.asarray(scale, dtype=np.float64), redshift=z, seed=seed,
                   beam=cube, field=field)
        out["conv_fft"] = FixtureBeam(box).convolve_fft(field)
        out["conv_fft_uniform"] = beams.BeamModel(box).convolve_fft(field)
        if with_real:                                   # O(N^4) per channel in the reference: the small case only
            out["conv_real"] = FixtureBeam(box).convolve_real(field)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print("wrote", name)


if __name__ == "__main__":
    main()
