"""GPU: the fused z pass of BASELINE configs[2] (k_rsd_turn: the inverse z transforms of delta and v_z, the
line-of-sight remap of box.py:384-438 and the forward z transform of the result in one kernel) against the separate
kernels it replaces -- the same instructions on the same numbers, so every product of the chain must agree bit for
bit: delta_x, v_z, the redshift-space field, the filtered field and the binned power spectrum.  (The separate kernels
are pinned to the oracle in tests/test_config3_gpu.py and tests/test_derived_gpu.py; the 512^3 chain of
test_config3_gpu.py runs through the fused pass and is held to the oracle there.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _box(N, seed=7, precision="f32"):
    from fastbox_amd import CosmoBox, default_cosmo
    return CosmoBox(cosmo=default_cosmo, box_scale=1e3 * N / 512., nsamp=N, realise_now=False, precision=precision,
                    rng="device", seed=seed)


def _chain(box, fused, method, sigma_nl, filtered=True):
    from fastbox_amd import Wedge
    from fastbox_amd.box import RedshiftSpaceField
    dx = box.realise_density()
    vz = box.to_real(box.realise_velocity()[2])
    if not fused:
        vz.ptr                                   # v_z finished on its own: the remap below runs as a kernel of its own
    ds = box.redshift_space_density(delta_x=dx, velocity_z=vz, sigma_nl=sigma_nl, method=method)
    assert isinstance(ds, RedshiftSpaceField) == fused
    if filtered:
        filt = box.apply_transfer_fn(box.to_k(ds), Wedge(slope=0.3))
        pend = box.binned_power_spectrum(delta_x=filt.real, nbins=20, wait=False)
        field = np.asarray(filt.real)
    else:
        pend = box.binned_power_spectrum(delta_x=ds, nbins=20, wait=False)
        field = None
    if fused:
        assert not ds.materialised and dx.materialised and not vz.materialised     # one pass did it all
    return pend.result(), np.asarray(dx), np.asarray(vz), np.asarray(ds), field


@pytest.mark.parametrize("method,sigma_nl", [("linear", 0.0), ("nearest", 0.0), ("linear", 150.0)])
@pytest.mark.parametrize("N", [64, 128, 256, 512])
def test_fused_z_pass_equals_separate_kernels(N, method, sigma_nl):
    a = _chain(_box(N), True, method, sigma_nl)
    b = _chain(_box(N), False, method, sigma_nl)
    for x, y, what in zip(a[1:], b[1:], ("delta_x", "v_z", "redshift-space field", "filtered field")):
        assert np.array_equal(x, y), what
    for x, y in zip(a[0], b[0]):
        assert np.array_equal(x, y, equal_nan=True)
    assert np.all(np.isfinite(a[4])) and np.std(a[3]) > 0.1 * np.std(a[1])


@pytest.mark.parametrize("N", [128, 512])
def test_fused_z_pass_without_a_filter(N):
    """binned_power_spectrum(delta_x=redshift_space_density(...)): the same pass, the binning x pass behind it.  The
    separate path bins a packed work spectrum (other partial sums): float32 rounding between the two."""
    a = _chain(_box(N, seed=3), True, "linear", 0.0, filtered=False)
    b = _chain(_box(N, seed=3), False, "linear", 0.0, filtered=False)
    for x, y in zip(a[1:4], b[1:4]):
        assert np.array_equal(x, y)
    m = ~np.isnan(b[0][1])
    assert np.array_equal(np.isnan(a[0][1]), np.isnan(b[0][1])) and np.array_equal(a[0][0], b[0][0])
    assert np.allclose(a[0][1][m], b[0][1][m], rtol=2e-6) and np.allclose(a[0][2][m], b[0][2][m], rtol=2e-5)


def test_fused_z_pass_is_not_taken_where_it_does_not_exist():
    """fp64 plans and fields that are already in memory go through the separate kernels."""
    from fastbox_amd.box import RedshiftSpaceField
    box = _box(64, precision="f64")
    dx = box.realise_density()
    vz = box.to_real(box.realise_velocity()[2])
    assert not isinstance(box.redshift_space_density(delta_x=dx, velocity_z=vz), RedshiftSpaceField)
    box = _box(64)
    dx = box.realise_density()
    vz = box.to_real(box.realise_velocity()[2])
    ds = box.redshift_space_density(delta_x=np.asarray(dx), velocity_z=vz)
    assert not isinstance(ds, RedshiftSpaceField)
    # a lazy field whose density was consumed by another estimate in between: finished the ordinary way
    dx = box.realise_density()
    vz = box.to_real(box.realise_velocity()[2])
    ds = box.redshift_space_density(delta_x=dx, velocity_z=vz)
    box.binned_power_spectrum(delta_x=dx, nbins=12)
    assert isinstance(ds, RedshiftSpaceField) and not ds.fusable()
    k, pk, err = box.binned_power_spectrum(delta_x=ds, nbins=12)
    assert np.all(np.isfinite(pk[~np.isnan(pk)])) and ds.materialised


def test_c_entry_point_refuses_what_it_was_not_built_for():
    """fb_power_spectrum_redshift_space: double-precision plans and grids above 512 get FB_ERR_UNSUPPORTED (the Python
    class then runs the separate kernels), identical work spectra and a missing bin table are argument / state errors."""
    from fastbox_amd import _lib
    from fastbox_amd.device import HALF
    for kw, N in ((dict(precision="f64"), 64), (dict(precision="f32"), 1024)):
        box = _box(N, **kw)
        eng = box.engine
        bins, kc, thr, amb = box._bin_setup(12, None)
        eng.set_bins(bins, thr, amb)
        a, b = eng.empty(HALF), eng.empty(HALF)
        with pytest.raises(_lib.FastBoxError) as err:
            eng.power_redshift_space(a, b, 70.0, 0.0, 1, "linear")
        assert "single-precision plan with 64 <= N <= 512" in str(err.value)
    box = _box(64)
    eng = box.engine
    bins, kc, thr, amb = box._bin_setup(12, None)
    eng.set_bins(bins, thr, amb)
    a = eng.empty(HALF)
    with pytest.raises(_lib.FastBoxError):
        eng.power_redshift_space(a, a, 70.0, 0.0, 1, "linear")            # one buffer for both fields
    with pytest.raises(_lib.FastBoxError):
        eng.power_redshift_space(a, eng.empty(HALF), -1.0, 0.0, 1, "linear")   # H(z) <= 0


def test_lazy_redshift_space_field_is_an_ordinary_field_to_its_users():
    """Whatever is done with the lazy field gives what the eager one gives: reading it (twice), arithmetic, the log-normal
    transform, an arbitrary transfer function on its transform, a second estimate after the fused one."""
    from fastbox_amd.box import RedshiftSpaceField

    def fields(fused):
        box = _box(64, seed=11)
        dx = box.realise_density()
        vz = box.to_real(box.realise_velocity()[2])
        if not fused:
            vz.ptr
        return box, dx, vz, box.redshift_space_density(delta_x=dx, velocity_z=vz)
    box, dx, vz, lazy = fields(True)
    _, _, _, eager = fields(False)
    assert isinstance(lazy, RedshiftSpaceField) and lazy.shape == eager.shape and lazy.dtype == eager.dtype
    want = np.asarray(eager)
    assert np.array_equal(np.asarray(lazy), want) and np.array_equal(np.asarray(lazy), want)
    assert np.array_equal(np.asarray(lazy + 1.0), want.astype(np.float32) + np.float32(1.0))      # (device arithmetic: float32)
    # the log-normal of a lazy field, and an arbitrary (host-evaluated) transfer function on its transform
    box, dx, vz, lazy = fields(True)
    box2, _, _, eager = fields(False)
    a = box.binned_power_spectrum(delta_x=box.lognormal(lazy), nbins=12)
    b = box2.binned_power_spectrum(delta_x=box2.lognormal(eager), nbins=12)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    box, dx, vz, lazy = fields(True)
    box2, _, _, eager = fields(False)
    tf = lambda kperp, kpar: np.exp(-0.5 * (kpar / 0.2) ** 2)
    fa = np.asarray(box.apply_transfer_fn(box.to_k(lazy), tf))
    fb = np.asarray(box2.apply_transfer_fn(box2.to_k(eager), tf))
    assert np.array_equal(fa, fb)
    # the fused estimate first, a plain one of the same field afterwards (the field is then formed the ordinary way)
    box, dx, vz, lazy = fields(True)
    box2, _, _, eager = fields(False)
    first = box.binned_power_spectrum(delta_x=lazy, nbins=12)
    assert not lazy.materialised
    np.asarray(lazy)
    again = box.binned_power_spectrum(delta_x=lazy, nbins=12)
    ref = box2.binned_power_spectrum(delta_x=eager, nbins=12)
    m = ~np.isnan(ref[1])
    assert np.allclose(first[1][m], ref[1][m], rtol=2e-6)
    for x, y in zip(again, ref):
        assert np.array_equal(x, y, equal_nan=True)
