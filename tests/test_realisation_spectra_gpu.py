"""GPU: CosmoBox.realisation_spectra / fb_montecarlo_power -- a Monte-Carlo loop queued by one library call -- against the
loop of realise_density() + [lognormal()] + binned_power_spectrum() it stands for (box.py:130-194, :441-460, :696-768):
identical numbers, realisation by realisation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _box(N, precision, L=1e3, seed=9, rng="device"):
    from fastbox_amd import CosmoBox, default_cosmo
    return CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, redshift=0., realise_now=False, precision=precision, rng=rng, seed=seed)


def _loop(box, indices, nbins, lognormal):
    rows = []
    for r in indices:
        box._realisation = r
        dx = box.realise_density()
        rows.append(box.binned_power_spectrum(delta_x=box.lognormal(dx) if lognormal else dx, nbins=nbins))
    return rows


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("N", [32, 128, 256])
def test_one_call_equals_the_loop(N, precision):
    for lognormal, nbins, stride in ((False, 20, 1), (True, 20, 3), (True, 50, 1)):
        a = _box(N, precision)
        a._realisation = 4
        kc, pk, sd = a.realisation_spectra(5, nbins=nbins, lognormal=lognormal, stride=stride)
        assert pk.shape == sd.shape == (5, nbins - 1) and a._realisation == 4 + 5 * stride
        want = _loop(_box(N, precision), [4 + i * stride for i in range(5)], nbins, lognormal)
        for i, (wk, wp, ws) in enumerate(want):
            assert np.array_equal(kc, wk)
            assert np.array_equal(pk[i], wp, equal_nan=True) and np.array_equal(sd[i], ws, equal_nan=True)
        assert not np.array_equal(pk[0], pk[1], equal_nan=True)
        kc2, pk2, _ = a.realisation_spectra(2, nbins=nbins, lognormal=lognormal, stride=stride)       # the counter moved on
        nxt = _loop(_box(N, precision), [4 + 5 * stride, 4 + 6 * stride], nbins, lognormal)
        assert np.array_equal(pk2[0], nxt[0][1], equal_nan=True) and np.array_equal(pk2[1], nxt[1][1], equal_nan=True)
    kc, pk, sd = _box(N, precision).realisation_spectra(0)
    assert pk.shape == (0, 19)


def test_boxes_without_the_fused_path_run_the_loop_and_the_host_stream_is_refused():
    # a cuboid (exact |k| per mode, no shell thresholds) and a grid that is not a power of two: same answers through the loop
    for N, L in ((32, (1e3, 2e3, 1.5e3)), (48, 1e3)):
        a = _box(N, "f64", L=L)
        kc, pk, sd = a.realisation_spectra(3, lognormal=True)
        want = _loop(_box(N, "f64", L=L), [0, 1, 2], 20, True)
        for i in range(3):
            assert np.array_equal(pk[i], want[i][1], equal_nan=True) and np.array_equal(sd[i], want[i][2], equal_nan=True)
    with pytest.raises(ValueError):
        _box(32, "f32", rng="numpy").realisation_spectra(2)
    with pytest.raises(ValueError):
        _box(32, "f32").realisation_spectra(-1)


def test_a_realisation_outside_the_shift_range_is_repeated_on_its_own():
    """0.5 Mpc voxels: sigma = 15, the variance-based shift of the exponentials fails for some realisations of a single-
    precision plan; those rows come from the repeat with the exact shift, as in binned_power_spectrum."""
    N = 64
    a = _box(N, "f32", L=32.)
    kc, pk, sd = a.realisation_spectra(6, lognormal=True)
    b = _box(N, "f32", L=32.)
    want = _loop(b, range(6), 20, True)
    for i in range(6):
        assert np.array_equal(pk[i], want[i][1], equal_nan=True)
    assert np.all(np.isfinite(pk[:, ~np.isnan(pk[0])]))
    assert a.lognormal_repeats == b.lognormal_repeats


def test_montecarlo_driver_uses_it_and_agrees_with_the_loop(tmp_path):
    from fastbox_amd import montecarlo
    a, b = _box(64, "f32"), _box(64, "f32")
    acc_a, kc_a, _ = montecarlo.run(a, 24, nbins=20, lognormal=True, batch=10, rank=1, world=3)
    acc_b, kc_b, _ = montecarlo.run(b, 24, nbins=20, lognormal=True, batch=10, rank=1, world=3, keep_fields=True)      # the loop
    assert acc_a.n == acc_b.n == 8 and np.array_equal(kc_a, kc_b)
    assert np.array_equal(acc_a.mean, acc_b.mean) and np.array_equal(acc_a.m2, acc_b.m2)
