"""BASELINE.json configs[2] at its stated size: the 512^3 chain

    realise_density -> realise_velocity[2] -> redshift_space_density -> apply_transfer_fn(Wedge) -> P(k) + filtered field

as bench.py's `config3` leg runs it (device generator, velocity field regenerated inside the generator pass, sort-free
remap with 8 cells per lane, filter + binning inside the forward transform's last pass), against oracle/box_oracle.py
-- the numpy restatement of the reference's algorithm (fastbox/box.py:130-194, 197-290, 356-438, 696-768) -- run on
the GPU host on the same inputs: the host model of the device noise (fastbox_amd/rng.py) coloured with the same P(k).

The remap is a bracket search, discontinuous in its inputs, so the chain is pinned on the fp64 plan (inputs agree to
1e-13: identical brackets); the fp32 plan (what the bench times) is then held to the fp64 plan's P(k).
"""
import numpy as np
import pytest

from oracle import box_oracle as bo
from oracle import standin

pytestmark = pytest.mark.gpu


def _chain(box, sigma_nl=0.0):
    from fastbox_amd import Wedge
    dx = box.realise_density()
    vz = box.to_real(box.realise_velocity()[2])
    ds = box.redshift_space_density(delta_x=dx, velocity_z=vz, sigma_nl=sigma_nl)
    filt = box.apply_transfer_fn(box.to_k(ds), Wedge(slope=0.3))
    pend = box.binned_power_spectrum(delta_x=filt.real, nbins=20, wait=False)
    return dx, vz, ds, filt, pend


def _close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b)) <= tol * np.sqrt(np.mean(np.abs(b) ** 2))


@pytest.mark.parametrize("N", [512])
def test_config3_chain_against_the_oracle(N):
    from fastbox_amd import CosmoBox, default_cosmo, rng
    L, seed = 1e3, 5
    box = CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, precision="f64", rng="device", seed=seed)
    dx, vz, ds, filt, pend = _chain(box)
    kc, pk, err = pend.result()
    got_dx, got_vz, got_ds = np.asarray(dx), np.asarray(vz), np.asarray(ds)
    got_f = np.asarray(filt.real)
    del dx, vz, ds, filt
    box.engine.release_idle_buffers()

    # ---- the oracle on the same noise
    geo = bo.box_geometry(L, N)
    cosmo = standin.cosmology()
    z = rng.half_spectrum_noise(N, seed, 0)
    k = bo.k_magnitude(geo)[:, :, :N // 2 + 1]
    amp = np.sqrt(np.nan_to_num(standin.pk_fn(cosmo, 1.0)(k.flatten())).reshape(k.shape) * geo["boxfactor"])
    z *= amp
    del k, amp
    want_dx = np.fft.irfftn(z, s=(N, N, N), axes=(0, 1, 2))
    del z
    assert _close(got_dx, want_dx, 1e-11)
    dk = np.fft.fftn(want_dx)
    want_vz = np.fft.ifftn(bo.realise_velocity(geo, dk, standin.velocity_fac(cosmo, 1.0))[2]).real
    del dk
    assert _close(got_vz, want_vz, 1e-10)
    want_ds = bo.redshift_space_density(geo, want_dx, want_vz, standin.hubble(cosmo, 1.0), 0.)
    assert _close(got_ds, want_ds, 1e-9)
    fw = bo.apply_transfer_fn(geo, np.fft.fftn(want_ds), standin.wedge03)
    assert _close(got_f, fw.real, 1e-9)
    okc, opk, oerr = bo.binned_power_spectrum(geo, np.fft.fftn(fw.real), nbins=20)
    m = ~np.isnan(opk)
    assert np.array_equal(np.isnan(pk), np.isnan(opk)) and np.array_equal(kc, okc)
    top = np.nanmax(opk)
    assert np.allclose(pk[m], opk[m], rtol=1e-9, atol=1e-12 * top) and np.allclose(err[m], oerr[m], rtol=1e-7, atol=1e-12 * top)

    # ---- the fp32 plan (the one bench.py times): the same realisation, P(k) of the filtered field against fp64.
    # A rounding-level change of a shifted coordinate can move a line-of-sight bracket, so isolated voxels differ;
    # the spectrum moves by far less than its own sampling error
    box32 = CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, precision="f32", rng="device", seed=seed)
    dx32, vz32, ds32, filt32, pend32 = _chain(box32)
    kc32, pk32, err32 = pend32.result()
    assert np.array_equal(kc32, okc) and np.array_equal(np.isnan(pk32), np.isnan(opk))
    assert np.allclose(pk32[m], opk[m], rtol=2e-4, atol=1e-7 * top)
    assert _close(dx32, want_dx, 3e-5) and _close(vz32, want_vz, 3e-5)
    d = np.abs(np.asarray(ds32) - want_ds)
    assert np.mean(d > 1e-3 * np.std(want_ds)) < 1e-3                 # fraction of voxels whose bracket moved
    # ... and pinned properly: the oracle's remap, filter and P(k) on the single-precision plan's OWN delta_x and v_z
    # (float64 arithmetic on the float32 fields).  What is left between the two is the device's arithmetic from the
    # remap on: a few brackets that flip on a last-bit difference of a shifted coordinate, and float32 rounding of the
    # transform -- the spectrum of the filtered field at the 1e-5 BASELINE states
    h_dx, h_vz = np.asarray(dx32), np.asarray(vz32)
    o_ds = bo.redshift_space_density(geo, h_dx, h_vz, standin.hubble(cosmo, 1.0), 0.)
    g_ds = np.asarray(ds32)
    flips = np.abs(g_ds - o_ds) > 1e-5 * np.max(np.abs(o_ds))
    assert flips.mean() < 2e-6, flips.sum()                            # a few hundred of 1.3e8 voxels at most
    o_fw = bo.apply_transfer_fn(geo, np.fft.fftn(o_ds), standin.wedge03)
    _, o_pk, o_err = bo.binned_power_spectrum(geo, np.fft.fftn(o_fw.real), nbins=20)
    assert np.allclose(pk32[m], o_pk[m], rtol=1e-5, atol=1e-9 * top), np.max(np.abs(pk32[m] / o_pk[m] - 1))
    assert _close(np.asarray(filt32.real), o_fw.real, 2e-5)


def test_redshift_space_eight_cells_per_lane_against_the_oracle():
    """N = 512: k_rsd_cells holds 8 cells per lane (tests elsewhere cover 1, 2 and 4).  Random fields with many wraps and
    crowded / empty cells, and the small-scale velocity noise of the device generator (stream 1), a sub-volume of
    lines of sight checked against the oracle."""
    from fastbox_amd import CosmoBox, default_cosmo, rng as hostrng
    N, seed = 512, 3
    box = CosmoBox(cosmo=default_cosmo, box_scale=3e2, nsamp=N, realise_now=False, precision="f64", rng="device", seed=seed)
    geo = bo.box_geometry(3e2, N)
    r = np.random.RandomState(5)
    d = r.normal(size=(N, N, N))
    v = 2e4 * r.normal(size=(N, N, N))
    Hz = standin.hubble(standin.cosmology(), 1.0)
    sub = slice(0, 24)                                              # 24 x 512 lines of sight through the oracle's loop

    class _Replay(object):
        def __init__(self, cube):
            self.lines = iter(cube.reshape(-1, N))

        def normal(self, loc, scale, size):
            return next(self.lines)
    got = np.asarray(box.redshift_space_density(delta_x=d, velocity_z=v, sigma_nl=0.))[sub]
    want = bo.redshift_space_density(geo, d[sub], v[sub], Hz, 0.)
    assert np.max(np.abs(got - want)) < 1e-12 * np.max(np.abs(want))
    los_seed = (box.seed + 0x9E3779B97F4A7C15 * (box._realisation + 1)) & (2 ** 64 - 1)
    got = np.asarray(box.redshift_space_density(delta_x=d, velocity_z=v, sigma_nl=150.))[sub]
    want = bo.redshift_space_density(geo, d[sub], v[sub], Hz, 150., _Replay(hostrng.los_noise(N, los_seed)[sub]))
    # libm vs device log / sin / cos differ in the last bits of the noise: a bracket may flip on a handful of cells
    assert np.mean(np.abs(got - want) > 1e-9 * np.max(np.abs(want))) < 1e-5


def test_redshift_space_single_precision_plan_at_512_against_the_oracle():
    """The kernel config 3 is timed on: k_rsd_cells<float, 8> (N = 512, 8 cells per lane, reciprocals instead of the
    two fp64 divisions, slopes in single precision).  Oracle in float64 on the SAME float32 inputs, 24 x 512 lines of
    sight with many wraps and crowded / empty cells: outputs agree to float32 rounding except where a last-bit
    difference of a shifted coordinate moves a bracket -- a handful of cells."""
    from fastbox_amd import CosmoBox, default_cosmo
    N, seed = 512, 3
    box = CosmoBox(cosmo=default_cosmo, box_scale=3e2, nsamp=N, realise_now=False, precision="f32", rng="device", seed=seed)
    geo = bo.box_geometry(3e2, N)
    r = np.random.RandomState(6)
    d = r.normal(size=(N, N, N)).astype(np.float32)
    v = (2e4 * r.normal(size=(N, N, N))).astype(np.float32)
    Hz = standin.hubble(standin.cosmology(), 1.0)
    sub = slice(100, 124)
    for method in ("linear", "nearest"):
        got = np.asarray(box.redshift_space_density(delta_x=d, velocity_z=v, sigma_nl=0., method=method))[sub]
        want = bo.redshift_space_density(geo, d[sub].astype(np.float64), v[sub].astype(np.float64), Hz, 0., method=method)
        bad = np.abs(got - want) > 2e-6 * np.max(np.abs(want)) + 2e-6 * np.abs(want)
        assert bad.sum() <= 12, (method, int(bad.sum()))               # of 6.3e6 cells
    # smooth, small displacements (config 3's regime: |v| / H ~ a few cells): no bracket is anywhere near flipping
    v2 = (3e2 * r.normal(size=(N, N, N))).astype(np.float32)
    got = np.asarray(box.redshift_space_density(delta_x=d, velocity_z=v2, sigma_nl=0.))[sub]
    want = bo.redshift_space_density(geo, d[sub].astype(np.float64), v2[sub].astype(np.float64), Hz, 0.)
    bad = np.abs(got - want) > 2e-6 * np.max(np.abs(want)) + 2e-6 * np.abs(want)
    assert bad.sum() <= 12, int(bad.sum())
