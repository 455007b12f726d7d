"""GPU: the log-normal transform (box.py:441-460) and its fused P(k) (box.py:696-768 on the transformed field) when the
field's variance is large -- the regime of BASELINE's own boxes (512^3 at 2 Mpc per voxel: sigma = 8; 2048^3 at 0.5 Mpc:
sigma = 21), where exp(delta) does not fit single precision and the device forms exp(delta - shift) instead
(fastbox_amd/hostgeom.py lognormal_shift).  The estimate exp(d)/<exp(d)> - 1 does not depend on the shift, so the oracle
is the reference's own arithmetic in float64 on the field the device drew."""
import gc

import numpy as np
import pytest

from oracle import box_oracle as bo
from oracle import standin

pytestmark = pytest.mark.gpu


def _steep_box(N, L, boost, precision="f32", **kw):
    """A CosmoBox whose input spectrum is the stand-in's times `boost` (sigma grows by sqrt(boost))."""
    from fastbox_amd import CosmoBox, default_cosmo

    class SteepBox(CosmoBox):
        def _power(self, k, scale_factor, linear):
            return boost * np.asarray(CosmoBox._power(self, k, scale_factor, linear))
    box = SteepBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, precision=precision, rng="device", **kw)
    box.box_scale_arg = L
    return box


def _oracle_ln_pk(box, dx_host, nbins=20):
    """(log-normal field, its binned P(k)) by the reference's float64 arithmetic on a host copy of the field."""
    ln = bo.lognormal(dx_host)
    return ln, bo.binned_power_spectrum(bo.box_geometry(box.box_scale_arg, box.N), np.fft.fftn(ln), nbins=nbins)


@pytest.mark.parametrize("boost,sigma_lo", [(1.0, 0.0), (60.0, 6.0), (900.0, 20.0)])
def test_fused_lognormal_spectrum_at_any_variance(boost, sigma_lo):
    """64^3, single-precision plan, sigma up to ~22 (a single voxel dominates <exp(d)>): fused P(k) of the log-normal
    field against the float64 oracle on the device's own delta_x, 1e-5; the materialised field likewise."""
    from fastbox_amd import hostgeom
    box = _steep_box(64, 1e3, boost, seed=21)
    dx = box.realise_density()
    pend = box.binned_power_spectrum(delta_x=box.lognormal(dx), nbins=20, wait=False)
    kc, pk, err = pend.result()
    d = np.asarray(dx)
    sigma = d.std()
    assert sigma > sigma_lo and abs(sigma ** 2 / box._sigma2 - 1) < 0.2
    ln, (okc, opk, oerr) = _oracle_ln_pk(box, d)
    m = ~np.isnan(opk)
    assert np.array_equal(np.isnan(pk), np.isnan(opk)) and np.array_equal(kc, okc)
    assert np.all(np.isfinite(pk[m])) and np.all(np.isfinite(err[m]))
    assert np.allclose(pk[m], opk[m], rtol=1e-5, atol=0), np.max(np.abs(pk[m] / opk[m] - 1))
    assert np.allclose(err[m], oerr[m], rtol=1e-4, atol=1e-5 * opk[m].max())
    assert not pend.repeated and box.lognormal_repeats == 0          # the analytic shift was good enough
    # the materialised field: exp(d - max d) / mean - 1, nothing overflows (exp(d) itself reaches 1e56 at sigma = 22)
    got = np.asarray(box.lognormal(dx))
    assert np.all(np.isfinite(got))
    assert np.max(np.abs(got - ln)) <= 3e-6 * np.max(np.abs(ln)) + 1e-6
    assert abs(got.mean()) < 1e-5 * max(1.0, np.abs(ln).max() / ln.size)


@pytest.mark.parametrize("bad_shift", [-400.0, 600.0])
def test_out_of_range_shift_is_detected_and_repeated(monkeypatch, bad_shift):
    """A shift that overflows (or flushes) every exponential: the non-finite (or empty) sums are caught when the
    spectrum is resolved, the step is repeated with the shift taken from the field's maximum, and the result is the
    one the good shift gives."""
    from fastbox_amd import hostgeom
    box = _steep_box(64, 1e3, 60.0, seed=22)
    want = [box.binned_power_spectrum(delta_x=box.lognormal(box.realise_density()), nbins=20) for _ in range(2)]
    box2 = _steep_box(64, 1e3, 60.0, seed=22)
    monkeypatch.setattr(hostgeom, "lognormal_shift", lambda sigma2, nvox: bad_shift)
    for i, keep in enumerate((True, False)):       # keep_field=False: the repeat draws the realisation again
        pend = box2.binned_power_spectrum(delta_x=box2.lognormal(box2.realise_density()), nbins=20, wait=False,
                                          keep_field=keep)
        got = pend.result()
        assert pend.repeated
        m = ~np.isnan(want[i][1])
        assert np.array_equal(np.isnan(got[1]), np.isnan(want[i][1]))
        assert np.allclose(got[1][m], want[i][1][m], rtol=2e-6) and np.allclose(got[2][m], want[i][2][m], rtol=2e-5)
    assert box2.lognormal_repeats == 2


def test_lognormal_of_a_foreign_field_takes_its_shift_from_the_data():
    """A field the box did not draw (host array, variance unknown to it): the maximum is reduced on the device first."""
    box = _steep_box(32, 5e2, 1.0, seed=1)
    rs = np.random.RandomState(5)
    field = 30.0 * rs.standard_normal((32, 32, 32))
    kc, pk, err = box.binned_power_spectrum(delta_x=box.lognormal(field), nbins=12)
    ln = bo.lognormal(field.astype(np.float32).astype(np.float64))
    okc, opk, oerr = bo.binned_power_spectrum(bo.box_geometry(5e2, 32), np.fft.fftn(ln), nbins=12)
    m = ~np.isnan(opk)
    assert np.array_equal(np.isnan(pk), np.isnan(opk))
    assert np.allclose(pk[m], opk[m], rtol=1e-5, atol=0)
    assert box.lognormal_repeats == 0
    assert box.engine.max_real(box.engine.upload(field, "real")) == np.float32(field.max())


@pytest.mark.parametrize("N", [1024, 2048])
def test_bench_box_lognormal_is_finite_at_1000_mpc(N):
    """The `sizes` leg of bench.py: N^3 at L = 1000 Mpc with the non-linear spectrum (sigma = 14 at 1024^3, 21 at
    2048^3).  Log-normal P(k) finite with the Gaussian field's NaN mask, both estimates bit-reproducible, the lazily
    materialised field finite with mean 0."""
    from fastbox_amd import CosmoBox, default_cosmo
    box = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f32", rng="device", seed=9)
    dx = box.realise_density()
    kc, pk, err = box.binned_power_spectrum(delta_x=box.lognormal(dx), nbins=20)
    gk, gpk, gerr = box.binned_power_spectrum(delta_x=dx, nbins=20)
    assert np.sqrt(box._sigma2) > (13.0 if N == 1024 else 20.0)
    assert np.array_equal(np.isnan(pk), np.isnan(gpk)) and np.array_equal(kc, gk)
    m = ~np.isnan(gpk)
    assert m.sum() >= 15 and np.all(np.isfinite(pk[m])) and np.all(pk[m] > 0) and np.all(np.isfinite(err[m]))
    assert box.lognormal_repeats == 0
    # exact shift (from the field's maximum) against the analytic one: the same estimate
    from fastbox_amd import hostgeom
    dmax = box.engine.max_real(dx)
    res, _ = box.engine.power_fused(dx, pre_exp=True, exp_shift=hostgeom.lognormal_shift_exact(dmax, float(N) ** 3))
    s1, s2, esum = box.engine.fetch_results(res, 20)
    mean = esum / float(N) ** 3
    pk2 = (s1 / mean ** 2 / (box.engine.bin_counts() * box.boxfactor))[1:]
    assert np.allclose(pk2[m], pk[m], rtol=2e-5)
    ln = box.lognormal(dx)
    tot = box.engine.sum_real(ln) / float(N) ** 3
    assert np.isfinite(tot) and abs(tot) < 1e-3                       # <exp(d)/mean - 1> = 0
    assert box.engine.max_real(ln) > 1e3                              # a handful of voxels carry the mean
    del ln, dx, box
    gc.collect()
