"""CPU: pins the numpy restatement (oracle/box_oracle.py) against the golden vectors that
oracle/make_golden.py captured from the reference itself (fastbox/box.py run here)."""
import os

import numpy as np
import pytest

from oracle import box_oracle as bo
from oracle import standin

CASES_ALL = ["n16_cube", "n16_cuboid", "n32_l1000", "n48_l1000", "n64_l1000", "n256_l1000"]
CASES_PK = ["n64_l100", "n64_l4000", "n128_l1000"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _scale(g):
    bs = g["box_scale"]
    return tuple(float(b) for b in bs) if bs.size == 3 else float(bs[0])


def _realise(g):
    """Regenerate the case from its seed with the oracle."""
    N = int(g["N"])
    geo = bo.box_geometry(_scale(g), N)
    a = 1. / (1. + float(g["redshift"]))
    cosmo = standin.cosmology()
    rng = np.random.RandomState(int(g["seed"]))
    re, im = bo.draw_noise(N, rng)
    dx, dk = bo.realise_density(geo, standin.pk_fn(cosmo, a), re, im)
    return geo, cosmo, a, rng, dx, dk


def _same(a, b):
    """Bit-for-bit (NaNs in the same places)."""
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("name", CASES_ALL + CASES_PK)
def test_geometry_density_pk(golden_dir, name):
    g = _load(golden_dir, name)
    geo, cosmo, a, rng, dx, dk = _realise(g)
    s = int(g["stride"])
    for key in ("Lx", "Ly", "Lz", "boxfactor", "kmin", "kmax"):
        assert geo[key] == float(g[key]), key
    assert _same(geo["x"], g["x"]) and _same(geo["z"], g["z"])
    assert _same(dx[::s, ::s, ::s], g["delta_x"])
    assert _same(dk[::s, ::s, ::s], g["delta_k"])
    assert np.sum(dx) == float(g["delta_x_sum"])
    for nb in (20, 50):
        kc, pk, err = bo.binned_power_spectrum(geo, dk, nbins=nb)
        assert _same(kc, g["pk%d_k" % nb]) and _same(pk, g["pk%d_p" % nb]) and _same(err, g["pk%d_e" % nb])
    kc, pk, err = bo.binned_power_spectrum(geo, dk, kbins=g["kbins"])
    assert _same(kc, g["pkkb_k"]) and _same(pk, g["pkkb_p"]) and _same(err, g["pkkb_e"])
    assert _same(np.array(bo.parseval(dx, dk)), g["parseval"])
    Hz = standin.hubble(cosmo, a)
    assert _same(bo.freq_array(geo, a, 1420.405752, Hz), g["freq_array"])
    # sigma_R from the realisation, the top-hat windows, the theory curve, test_sampling_error's numbers
    h = cosmo['h']
    assert bo.sigma_R(geo, dk, 8., h) == float(g["sigma8"]) and bo.sigma_R(geo, dk, 20., h) == float(g["sigmaR20"])
    assert _same(bo.tophat_window(g["window_k"], 8. / h), g["window8"])
    assert _same(bo.tophat_window1(g["window_k"], 8. / h), g["window1_8"])
    tk = np.logspace(-3.5, 1., int(1e3))
    assert _same(tk[::25], g["theory_k"]) and _same(standin.pk_fn(cosmo, a)(tk)[::25], g["theory_pk"])
    if "sampling_report" in g.files:
        rep = bo.sampling_report(geo, dx, dk, standin.pk_fn(cosmo, a), h)
        assert np.allclose(rep, g["sampling_report"], rtol=1e-12, atol=0)     # (parsed from 17-digit prints)


@pytest.mark.parametrize("name", CASES_ALL)
def test_derived_fields(golden_dir, name):
    g = _load(golden_dir, name)
    geo, cosmo, a, rng, dx, dk = _realise(g)
    s = int(g["stride"])
    p = lambda x: x[::s, ::s, ::s]
    ln = bo.lognormal(dx)
    assert _same(p(ln), g["lognormal"])
    kc, pk, err = bo.binned_power_spectrum(geo, np.fft.fftn(ln))
    assert _same(pk, g["pkln_p"]) and _same(err, g["pkln_e"])
    assert _same(p(bo.apply_transfer_fn(geo, dk, standin.beam_highpass)), g["tf_beam"])
    assert _same(p(bo.apply_transfer_fn(geo, dk, standin.highpass3)), g["tf_hp3"])
    assert _same(p(bo.apply_transfer_fn(geo, dk, standin.wedge03)), g["tf_wedge"])
    assert _same(p(bo.smooth_field(geo, dk, 8.0, cosmo['h'])), g["smooth8"])
    vel = bo.realise_velocity(geo, dk, standin.velocity_fac(cosmo, a))
    for c in range(3):
        assert _same(p(vel[c]), g["vel%d_k" % c])
    vz = np.fft.ifftn(vel[2]).real
    assert _same(p(vz), g["vel_z"])
    assert _same(p(bo.realise_potential(geo, dk)), g["phi_k"])
    Hz = standin.hubble(cosmo, 1. / (1. + float(g["redshift"])))
    rsd0 = bo.redshift_space_density(geo, dx, vz, Hz, 0.)
    assert _same(p(rsd0), g["rsd0"])
    assert _same(p(bo.redshift_space_density(geo, dx, vz, Hz, 200., rng)), g["rsd200"])
    if "rsd0_cubic" in g.files:      # griddata's third 1-D rule: the restatement builds the same not-a-knot spline by another scipy route
        cub = p(bo.redshift_space_density(geo, dx, vz, Hz, 0., method='cubic'))
        assert np.max(np.abs(cub - g["rsd0_cubic"])) <= 1e-11 * np.max(np.abs(g["rsd0_cubic"]))
    kc, pk, err = bo.binned_power_spectrum(geo, np.fft.fftn(rsd0))
    assert _same(pk, g["pkrsd_p"])
    # BASELINE configs[2]: wedge-filtered redshift-space field and its P(k)
    fw = bo.apply_transfer_fn(geo, np.fft.fftn(rsd0), standin.wedge03)
    assert _same(p(fw), g["rsd_wedge"])
    kc, pk, err = bo.binned_power_spectrum(geo, np.fft.fftn(fw.real))
    assert _same(kc, g["pkrsdw_k"]) and _same(pk, g["pkrsdw_p"]) and _same(err, g["pkrsdw_e"])
    assert _same(p(bo.redshift_space_density(geo, dx, vz, Hz, 0., method='nearest')), g["rsd0_nearest"])


@pytest.mark.skipif(os.environ.get("FASTBOX_SLOW_TESTS", "0") != "1",
                    reason="~3 min and ~20 GB of host work: set FASTBOX_SLOW_TESTS=1")
def test_oracle_at_the_headline_size(golden_dir):
    """BASELINE.json configs[1]: the oracle against what the reference produced at 512^3."""
    g = _load(golden_dir, "n512_l1000")
    geo, cosmo, a, rng, dx, dk = _realise(g)
    s = int(g["stride"])
    assert _same(dx[::s, ::s, ::s], g["delta_x"]) and _same(dk[::s, ::s, ::s], g["delta_k"])
    assert np.sum(dx) == float(g["delta_x_sum"])
    kc, pk, err = bo.binned_power_spectrum(geo, dk, nbins=20)
    assert _same(kc, g["pk20_k"]) and _same(pk, g["pk20_p"]) and _same(err, g["pk20_e"])
    ln = bo.lognormal(dx)
    assert _same(ln[::s, ::s, ::s], g["lognormal"])
    kc, pk, err = bo.binned_power_spectrum(geo, np.fft.fftn(ln))
    assert _same(pk, g["pkln_p"]) and _same(err, g["pkln_e"])


def test_oracle_against_live_reference():
    """When the reference is on this machine, compare directly on a fresh seed."""
    from oracle import ref_loader
    if not ref_loader.reference_available():
        pytest.skip("reference sources not present (GPU box)")
    ref = ref_loader.load_reference_box()
    np.random.seed(5)
    box = ref.CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=(3e2, 3e2, 3e2), nsamp=24, realise_now=False)
    box.realise_density()
    geo = bo.box_geometry((3e2, 3e2, 3e2), 24)
    rng = np.random.RandomState(5)
    re, im = bo.draw_noise(24, rng)
    dx, dk = bo.realise_density(geo, standin.pk_fn(standin.cosmology(), 1.0), re, im)
    assert _same(dx, box.delta_x) and _same(dk, box.delta_k)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref_pk = box.binned_power_spectrum(nbins=30)
    ora_pk = bo.binned_power_spectrum(geo, dk, nbins=30)
    for a, b in zip(ref_pk, ora_pk):
        assert _same(a, b)


# ---- noise / foreground steps: oracle/sky_oracle.py against vectors captured from the reference
@pytest.mark.parametrize("name", ["sky_n16", "sky_n32"])
def test_sky_oracle_reproduces_reference_vectors(golden_dir, name):
    from oracle import sky_oracle as so
    from fastbox_amd import cosmology
    g = _load(golden_dir, name)
    N, z, seed = int(g["N"]), float(g["redshift"]), int(g["seed"])
    geo = bo.box_geometry(tuple(g["box_scale"]), N)
    a = 1. / (1. + z)
    cosmo = standin.cosmology()
    freqs = bo.freq_array(geo, a, 1420.405752, standin.hubble(cosmo, a))
    r = cosmology.comoving_angular_distance(cosmo, a)
    ang_x, ang_y = bo.pixel_array(geo, r)
    assert _same(freqs, g["freqs"]) and _same(ang_x, g["ang_x"])
    pix = ang_x[1] - ang_x[0]
    rng = np.random.RandomState(seed + 1)
    fg_map = so.foreground_amp(geo, r, 57., 1.1, 10., sigma_pix=4. / pix, rng=rng)
    alpha = so.spectral_index(N, 2.07, 0.0002, 15. / pix, rng=rng)
    assert _same(fg_map, g["fg_map"]) and _same(alpha, g["alpha"])
    assert _same(so.construct_cube(fg_map, alpha, freqs, 130.), g["fg_cube"])
    raw = so.foreground_amp(geo, r, 57., 1.1, 10., rng=np.random.RandomState(seed + 2))
    assert _same(raw, g["fg_map_raw"]) and _same(so.construct_cube(raw, 2.1, freqs, 130.), g["fg_cube_scalar"])
    sig = so.radiometer_sigma(freqs, ang_x, 18., 2., 1., 64)
    assert _same(so.radiometer_noise((N, N, N), sig, np.random.RandomState(seed + 3)), g["noise_cube"])
    # the separable kernel the device smoothing uses is scipy's
    w, rad = so.gaussian_weights(4. / pix)
    import scipy.ndimage
    x = np.random.RandomState(1).normal(size=(N, N))
    mine = x
    for ax in (0, 1):
        mine = sum(w[j] * np.roll(mine, rad - j, axis=ax) for j in range(2 * rad + 1))
    assert np.max(np.abs(mine - scipy.ndimage.gaussian_filter(x, 4. / pix, mode="wrap"))) < 1e-13


@pytest.mark.parametrize("name", ["pca_n16", "pca_n32"])
def test_pca_oracle_reproduces_reference_vectors(golden_dir, name):
    from oracle import pca_oracle as po
    g = _load(golden_dir, name)
    data = g["data"]
    assert _same(po.mean_spectrum_filter(data), g["mean_sub"])
    assert _same(po.angular_bandpass_filter(data, 0.08, 0.3, d=1.), g["bandpass"])
    assert _same(po.angular_bandpass_filter(data, 0.0, 0.11, d=2.), g["bandpass_d2"])
    for nm in (2, 4):
        cleaned, U, amps = po.pca_filter(data, nm, return_filter=True)
        assert _same(cleaned, g["cleaned%d" % nm]) and _same(U, g["U%d" % nm]) and _same(amps, g["amps%d" % nm])
        cpl, Upl, apl = po.pca_filter(data, nm, fit_powerlaw=True, return_filter=True)
        assert _same(cpl, g["cleaned_pl%d" % nm]) and _same(Upl, g["U_pl%d" % nm]) and _same(apl, g["amps_pl%d" % nm])
        # what the device path is held to: the cleaned cube only depends on the span of the leading modes
        _, x, cov = po.channel_covariance(data)
        w, v = np.linalg.eigh(cov)
        Uh = v[:, ::-1][:, :nm]
        alt = data - (Uh @ (Uh.T @ x) + np.mean(data.reshape(-1, data.shape[-1]), axis=0)[:, None]).T.reshape(data.shape)
        assert np.max(np.abs(alt - cleaned)) < 1e-9 * np.max(np.abs(data))


def test_host_side_helpers_of_the_sky_models_match_the_oracle():
    """Host logic that needs no GPU: the smoothing kernel weights handed to fb_sky_gaussian_filter are scipy's."""
    from fastbox_amd import sky
    from oracle import sky_oracle as so
    for sigma in (0.3, 1.0, 2.7, 11.5):
        w, r = sky._gaussian_weights(sigma)
        wo, ro = so.gaussian_weights(sigma)
        assert r == ro and np.array_equal(w, wo) and abs(w.sum() - 1) < 1e-15


@pytest.mark.parametrize("name", ["beam_n16", "beam_n32"])
def test_beam_oracle_reproduces_reference_vectors(golden_dir, name):
    """oracle/beam_oracle.py against BeamModel.convolve_fft / convolve_real run by the reference itself
    (oracle/make_golden_beams.py).  scipy's fftconvolve uses real transforms of another length and convolve2d a
    direct sum, so agreement is to rounding (1e-12 of the field's scale), not bit for bit."""
    from oracle import beam_oracle as bo_beam
    g = _load(golden_dir, name)
    beam, field = g["beam"], g["field"]
    scale = np.max(np.abs(g["conv_fft"]))
    assert np.max(np.abs(bo_beam.convolve_fft(beam, field) - g["conv_fft"])) < 1e-12 * scale
    assert np.max(np.abs(bo_beam.convolve_fft(np.ones_like(beam), field) - g["conv_fft_uniform"])) < 1e-12 * scale
    if "conv_real" in g.files:
        assert np.max(np.abs(bo_beam.convolve_real_direct(beam, field) - g["conv_real"])) < 1e-12 * scale
        assert np.max(np.abs(bo_beam.convolve_real(beam, field) - g["conv_real"])) < 1e-12 * scale
    # the fixture's beam is an input produced by the oracle module: it must regenerate from the stored grid
    assert g["beam"].shape == field.shape
