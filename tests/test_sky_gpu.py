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


@pytest.mark.parametrize("name", ["pca_n16", "pca_n32"])
@pytest.mark.parametrize("precision,tol", [("f64", 1e-9), ("f32", 3e-6)])
@pytest.mark.parametrize("eigensolver", ["host", "device"])
def test_pca_filter_matches_reference_vectors(golden_dir, name, precision, tol, eigensolver):
    """fastbox_amd.filters against vectors from the reference's filters.py: the cleaned cube (which depends only on
    the span of the modes), the projector U U^T, the amplitudes up to a sign per mode -- with the modes from LAPACK on the host and from
    fb_leading_eigenvectors on the device."""
    from fastbox_amd import CosmoBox, filters
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    N, data = int(g["N"]), g["data"]
    box = CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=1e3, nsamp=N, realise_now=False, precision=precision)
    cube = box.engine.upload(data, "real")
    scale = np.max(np.abs(data))
    ms = np.asarray(filters.mean_spectrum_filter(cube))
    assert np.max(np.abs(ms - g["mean_sub"])) < tol * scale
    for nm in (2, 4):
        cleaned, U, amps = filters.pca_filter(cube, nm, return_filter=True, eigensolver=eigensolver)
        assert np.max(np.abs(np.asarray(cleaned) - g["cleaned%d" % nm])) < tol * scale
        Ur = g["U%d" % nm].real
        # fp32 storage of the cube perturbs the noise-dominated modes (nearly degenerate eigenvalues) at the 1e-4
        # level; the cleaned cube above is insensitive to that
        assert U.shape == Ur.shape and np.max(np.abs(U @ U.T - Ur @ Ur.T)) < (1e-3 if precision == "f32" else 1e-9)
        sgn = np.sign(np.sum(U * Ur, axis=0))
        ref_amps = g["amps%d" % nm].real
        assert np.max(np.abs(amps * sgn[:, None] - ref_amps)) < (2e-3 if precision == "f32" else 1e-9) * np.max(np.abs(ref_amps))
        only = filters.pca_filter(data, nm, box=box, eigensolver=eigensolver)                       # host array in, no filter returned
        assert np.max(np.abs(np.asarray(only) - g["cleaned%d" % nm])) < tol * scale
        # fit_powerlaw=True: the modes come from the covariance about the TRUE channel means (np.cov re-centres),
        # only the subtracted / restored spectrum is the power-law fit (filters.py:146-158)
        cleaned, U, amps = filters.pca_filter(cube, nm, fit_powerlaw=True, return_filter=True, eigensolver=eigensolver)
        # (fp32 storage of the cube moves the channel means by 1e-7, and the least-squares power-law fit of them -- which
        # stays in the cleaned cube, since it is not the true mean spectrum -- answers with 5e-5)
        assert np.max(np.abs(np.asarray(cleaned) - g["cleaned_pl%d" % nm])) < (tol if precision == "f64" else 2e-4) * scale
        Ur = g["U_pl%d" % nm].real
        assert np.max(np.abs(U @ U.T - Ur @ Ur.T)) < (1e-3 if precision == "f32" else 1e-9)
        sgn = np.sign(np.sum(U * Ur, axis=0))
        ref_amps = g["amps_pl%d" % nm].real
        # (the amplitudes carry the fitted spectrum, and scipy's curve_fit stops at a relative 1.5e-8: fed channel means
        # that differ in the last bits it lands a few 1e-9 away; the cleaned cube above is insensitive to that)
        assert np.max(np.abs(amps * sgn[:, None] - ref_amps)) < (1e-2 if precision == "f32" else 1e-7) * np.max(np.abs(ref_amps))


def test_channel_covariance_on_the_matrix_cores_full_size():
    """N = 256 (the 64 x 64 block shape, several block pairs and pixel slices): the MFMA covariance against numpy
    on the same fp32 cube, and the cleaning identities (cleaned channels have zero mean and no component along U)."""
    from fastbox_amd import CosmoBox, filters, _lib
    import ctypes
    N = 256
    box = CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=1e3, nsamp=N, realise_now=False, precision="f32", rng="device", seed=3)
    rs = np.random.RandomState(0)
    spec = 50. * (np.linspace(1., 2., N)[None, :] ** -2.7) * (1. + 0.3 * rs.normal(size=(N * N, 1)))
    data = (spec + 0.05 * rs.normal(size=(N * N, N))).astype(np.float32).reshape(N, N, N)
    cube = box.engine.upload(data, "real")
    eng = box.engine
    mean = filters._channel_means(eng, cube)
    cov_dev = eng._alloc_bytes(N * N * 8)
    _lib.call("fb_channel_covariance", eng._plan, cube.ptr, mean.ptr, cov_dev.ptr, eng.stream)
    cov = np.empty((N, N))
    _lib.call("fb_memcpy_d2h", cov.ctypes.data_as(ctypes.c_void_p), cov_dev.ptr, cov.nbytes, eng.stream)
    want = np.cov(data.reshape(-1, N).astype(np.float64).T)
    assert np.max(np.abs(cov - want)) < 1e-11 * np.max(np.abs(want))
    assert np.array_equal(cov, cov.T)
    cleaned, U, amps = filters.pca_filter(cube, 3, return_filter=True)
    c = np.asarray(cleaned).reshape(-1, N)
    assert np.max(np.abs(c.mean(axis=0))) < 1e-6 and np.max(np.abs(c @ U)) < 2e-4
    assert np.std(c) < 0.06                                                # the smooth foreground is gone


@pytest.mark.parametrize("name", ["pca_n16", "pca_n32"])
@pytest.mark.parametrize("precision,tol", [("f64", 1e-12), ("f32", 2e-6)])
def test_angular_bandpass_filter_matches_reference_vectors(golden_dir, name, precision, tol):
    from fastbox_amd import CosmoBox, filters
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    N, data = int(g["N"]), g["data"]
    box = CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=1e3, nsamp=N, realise_now=False, precision=precision)
    scale = np.max(np.abs(data))
    got = np.asarray(filters.angular_bandpass_filter(box.engine.upload(data, "real"), 0.08, 0.3, d=1.))
    assert got.dtype == np.complex128 and np.max(np.abs(got - g["bandpass"])) < tol * scale
    got2 = np.asarray(filters.angular_bandpass_filter(data, 0.0, 0.11, d=2., box=box))
    assert np.max(np.abs(got2 - g["bandpass_d2"])) < tol * scale
