"""
TEST INFRASTRUCTURE ONLY.  Golden vectors for the noise / foreground steps, captured from the reference's own
fastbox/noise.py and fastbox/foregrounds.py (loaded by path, stand-in pyccl) on top of the reference CosmoBox.
Run in the build container only:   python -m oracle.make_golden_sky
Fixtures hold seeds, parameters and expected outputs, never reference source.
"""
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import standin                                              # noqa: E402
from oracle.ref_loader import load_reference_box, load_reference_module, load_reference_filters   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CASES = [("sky_n16", 16, (3e3, 3e3, 3e3), 0.8, 5), ("sky_n32", 32, (4e3, 4e3, 2e3), 0.5, 9)]


def main():
    ref = load_reference_box()
    noise_mod = load_reference_module("fastbox/noise.py", "_fastbox_reference_noise")
    fg_mod = load_reference_module("fastbox/foregrounds.py", "_fastbox_reference_foregrounds")
    for name, N, scale, z, seed in CASES:
        np.random.seed(seed)
        box = ref.CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=scale, nsamp=N, redshift=z, realise_now=False)
        out = dict(numpy_version=np.__version__, scipy_version=scipy.__version__, N=N,
                   box_scale=np.asarray(scale, dtype=np.float64), redshift=z, seed=seed)
        out["freqs"] = box.freq_array()
        ang_x, ang_y = box.pixel_array()
        out["ang_x"] = ang_x
        # foregrounds.py (example_endtoend.py:58-72 parameters); draws: amp map (re, im), then spectral index
        fg = fg_mod.ForegroundModel(box)
        np.random.seed(seed + 1)
        out["fg_map"] = fg.realise_foreground_amp(amp=57., beta=1.1, monopole=10., smoothing_scale=4., redshift=z)
        out["alpha"] = fg.realise_spectral_index(mean_spec_idx=2.07, std_spec_idx=0.0002, smoothing_scale=15.,
                                                 redshift=z)
        out["fg_cube"] = fg.construct_cube(out["fg_map"], out["alpha"], freq_ref=130., redshift=z)
        np.random.seed(seed + 2)
        out["fg_map_raw"] = fg.realise_foreground_amp(amp=57., beta=1.1, monopole=10., smoothing_scale=None)
        out["fg_cube_scalar"] = fg.construct_cube(out["fg_map_raw"], 2.1, freq_ref=130.)
        # noise.py (example_endtoend.py:81-83 parameters)
        np.random.seed(seed + 3)
        out["noise_cube"] = noise_mod.NoiseModel(box).realise_radiometer_noise(Tinst=18., tp=2., fov=1., Ndish=64)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        # filters.py on a data cube = signal-like field + foregrounds + noise (example_endtoend.py:105-111)
        filt = load_reference_filters()
        rs = np.random.RandomState(seed + 4)
        data = 0.1 * (1. + 0.3 * rs.normal(size=(N, N, N))) + out["fg_cube"] + out["noise_cube"]
        pca = dict(N=N, seed=seed, data=data, mean_sub=filt.mean_spectrum_filter(data))
        for nm in (2, 4):
            cleaned, U, amps = filt.pca_filter(data, nmodes=nm, return_filter=True)
            pca["cleaned%d" % nm] = cleaned
            pca["U%d" % nm] = U
            pca["amps%d" % nm] = amps
            # fit_powerlaw=True (filters.py:146-154): modes projected out of d - (power-law fit of the mean spectrum)
            cleaned, U, amps = filt.pca_filter(data, nmodes=nm, fit_powerlaw=True, return_filter=True)
            pca["cleaned_pl%d" % nm], pca["U_pl%d" % nm], pca["amps_pl%d" % nm] = cleaned, U, amps
        pca["bandpass"] = filt.angular_bandpass_filter(data, 0.08, 0.3, d=1.)
        pca["bandpass_d2"] = filt.angular_bandpass_filter(data, 0.0, 0.11, d=2.)
        np.savez_compressed(os.path.join(OUT, name.replace("sky", "pca") + ".npz"), **pca)
        print("wrote", name, {k: getattr(v, "shape", v) for k, v in out.items() if k in ("fg_map", "fg_cube", "noise_cube")})


if __name__ == "__main__":
    main()
