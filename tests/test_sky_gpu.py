"""Foreground model and radiometer noise on the device (fastbox_amd/sky.py) against vectors captured from the
reference's fastbox/foregrounds.py and fastbox/noise.py (tests/golden/sky_*.npz, oracle/make_golden_sky.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import sky_oracle as so        # noqa: E402
from oracle import standin                  # noqa: E402


def _case(golden_dir, name, precision, rng="numpy", seed=0):
    from fastbox_amd import CosmoBox
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    box = CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=tuple(g["box_scale"]), nsamp=int(g["N"]),
                   redshift=float(g["redshift"]), realise_now=False, precision=precision, rng=rng, seed=seed)
    return g, box


@pytest.mark.parametrize("name", ["sky_n16", "sky_n32"])
@pytest.mark.parametrize("precision,tol", [("f64", 1e-12), ("f32", 2e-6)])
def test_foregrounds_and_noise_match_reference_vectors(golden_dir, name, precision, tol):
    from fastbox_amd import ForegroundModel, NoiseModel
    g, box = _case(golden_dir, name, precision)
    seed, z = int(g["seed"]), float(g["redshift"])
    assert np.allclose(box.freq_array(), g["freqs"], rtol=1e-14)
    fg = ForegroundModel(box)
    np.random.seed(seed + 1)                                         # the draw order of the capture script
    fg_map = fg.realise_foreground_amp(amp=57., beta=1.1, monopole=10., smoothing_scale=4., redshift=z)
    alpha = fg.realise_spectral_index(mean_spec_idx=2.07, std_spec_idx=0.0002, smoothing_scale=15., redshift=z)
    assert fg_map.shape == (box.N, box.N) and fg_map.dtype == np.float64
    scale = np.max(np.abs(g["fg_map"] - 10.))
    assert np.max(np.abs(fg_map - g["fg_map"])) < 40 * tol * max(scale, 1.)
    assert np.max(np.abs(alpha - g["alpha"])) < tol * 3
    cube = np.asarray(fg.construct_cube(g["fg_map"], g["alpha"], freq_ref=130., redshift=z))
    assert np.max(np.abs(cube / g["fg_cube"] - 1)) < max(tol, 3e-6 if precision == "f32" else 0)
    np.random.seed(seed + 2)
    raw = fg.realise_foreground_amp(amp=57., beta=1.1, monopole=10., smoothing_scale=None)
    assert np.max(np.abs(raw - g["fg_map_raw"])) < 40 * tol * np.max(np.abs(g["fg_map_raw"] - 10.))
    cube2 = np.asarray(fg.construct_cube(g["fg_map_raw"], 2.1, freq_ref=130.))
    assert np.max(np.abs(cube2 / g["fg_cube_scalar"] - 1)) < max(tol, 3e-6 if precision == "f32" else 0)
    np.random.seed(seed + 3)
    noise = np.asarray(NoiseModel(box).realise_radiometer_noise(Tinst=18., tp=2., fov=1., Ndish=64))
    assert np.max(np.abs(noise - g["noise_cube"])) < tol * np.max(np.abs(g["noise_cube"]))


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_device_generator_sky_statistics(precision):
    """rng='device': per-channel rms of the noise cube = the radiometer sigma, foreground map mean = monopole with
    the spectrum of the model, reproducible per seed."""
    from fastbox_amd import CosmoBox, ForegroundModel, NoiseModel
    N = 128
    box = CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=2e3, nsamp=N, redshift=0.8, realise_now=False,
                   precision=precision, rng="device", seed=4)
    ang_x, _ = box.pixel_array()
    sigma = so.radiometer_sigma(box.freq_array(), ang_x, 18., 2., 1., 64)
    nm = NoiseModel(box)
    cube = np.asarray(nm.realise_radiometer_noise(Tinst=18., tp=2., fov=1., Ndish=64))
    rms = np.sqrt(np.mean(cube ** 2, axis=(0, 1)))
    assert np.max(np.abs(rms / sigma - 1)) < 5 / np.sqrt(2. * N * N)          # 5 sigma of a chi^2 with N^2 dof
    assert abs(np.mean(cube / sigma[None, None, :])) < 5 / np.sqrt(float(N) ** 3)
    cube_b = np.asarray(nm.realise_radiometer_noise(Tinst=18., tp=2., fov=1., Ndish=64))
    assert np.abs(np.corrcoef(cube.ravel(), cube_b.ravel())[0, 1]) < 5 / np.sqrt(float(N) ** 3)   # a fresh draw
    fg = ForegroundModel(box)
    m = fg.realise_foreground_amp(amp=57., beta=-2.7, monopole=10.)
    assert abs(np.mean(m) - 10.) < 1e-4                                       # zero mode removed, monopole added
    # power of the map follows C_ell: compare low-k and high-k band powers with the model's ratio
    from fastbox_amd import cosmology
    r = cosmology.comoving_angular_distance(box.cosmo, box.scale_factor)
    geo = dict(N=N, Lx=box.Lx, Ly=box.Ly)
    k_perp, C = so.foreground_cell(geo, r, 57., -2.7)
    pk = np.abs(np.fft.fftn(m - 10.)) ** 2
    lo = (k_perp > 0) & (k_perp < 4 * k_perp[0, 1]); hi = (k_perp > 16 * k_perp[0, 1]) & (k_perp < 24 * k_perp[0, 1])
    got, want = np.mean(pk[lo]) / np.mean(pk[hi]), np.mean(C[lo]) / np.mean(C[hi])
    assert abs(got / want - 1) < 0.35                                         # 44 modes in the low band
    box2 = CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=2e3, nsamp=N, redshift=0.8, realise_now=False,
                    precision=precision, rng="device", seed=4)
    assert np.array_equal(np.asarray(NoiseModel(box2).realise_radiometer_noise(Tinst=18., tp=2., fov=1., Ndish=64)), cube)


@pytest.mark.parametrize("precision,tol", [("f32", 1e-6), ("f64", 1e-14)])
def test_cube_arithmetic_stays_on_the_device(precision, tol):
    """The numpy expressions callers write between the steps (example_endtoend.py:47, :75, :86) on DeviceArrays."""
    from fastbox_amd import CosmoBox, DeviceArray
    N = 32
    box = CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=1e3, nsamp=N, realise_now=False, precision=precision,
                   rng="device", seed=1)
    a = box.realise_density()
    b = box.lognormal(box.realise_density())
    ha, hb = np.asarray(a), np.asarray(b)
    cases = [(a + b, ha + hb), (a - b, ha - hb), (a * b, ha * hb), (0.3 * (1. + a), 0.3 * (1. + ha)), (a * 2, ha * 2),
             (2. - a, 2. - ha), (a - 2., ha - 2.), (-a, -ha)]
    for got, want in cases:
        assert isinstance(got, DeviceArray) and got.kind == "real"
        assert np.max(np.abs(np.asarray(got) - want)) <= tol * np.max(np.abs(want))
    assert isinstance(a + ha, np.ndarray)                       # mixed with a host array: numpy semantics
    cube = a
    cube += b                                                   # __iadd__ falls back to a new device cube
    assert isinstance(cube, DeviceArray) and np.max(np.abs(np.asarray(cube) - (ha + hb))) <= tol * np.max(np.abs(ha + hb))
