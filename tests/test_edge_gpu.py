"""GPU: edge cases of the reference interface and size-independent properties at BASELINE sizes."""
import numpy as np
import pytest

from oracle import box_oracle as bo
from oracle import standin

pytestmark = pytest.mark.gpu


def _box(N=32, L=5e2, **kw):
    from fastbox_amd import CosmoBox, default_cosmo
    return CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, **kw)


@pytest.mark.parametrize("precision,tol", [("f32", 1e-5), ("f64", 1e-11)])
def test_host_arrays_and_non_hermitian_spectra(precision, tol):
    """Host ndarrays go in as the reference accepts them: real fields, and arbitrary complex
    (non-Hermitian) spectra, which take the full-layout kernels (binning, filter, velocity, c2c)."""
    N, L = 32, 5e2
    box = _box(N, L, precision=precision)
    geo = bo.box_geometry(L, N)
    rng = np.random.RandomState(2)
    fk = rng.normal(size=(N, N, N)) + 1j * rng.normal(size=(N, N, N))
    fx = rng.normal(size=(N, N, N))
    got = box.binned_power_spectrum(delta_k=fk, nbins=15)
    want = bo.binned_power_spectrum(geo, fk, nbins=15)
    m = ~np.isnan(want[1])
    assert np.array_equal(got[0], want[0]) and np.array_equal(np.isnan(got[1]), np.isnan(want[1]))
    assert np.allclose(got[1][m], want[1][m], rtol=10 * tol) and np.allclose(got[2][m], want[2][m], rtol=10 * tol)
    got = box.binned_power_spectrum(delta_x=fx, nbins=15)
    want = bo.binned_power_spectrum(geo, np.fft.fftn(fx), nbins=15)
    assert np.allclose(got[1][m], want[1][m], rtol=10 * tol)
    out = np.asarray(box.apply_transfer_fn(fk, standin.beam_highpass))
    want = bo.apply_transfer_fn(geo, fk, standin.beam_highpass)
    assert np.max(np.abs(out - want)) < 10 * tol * np.std(want)
    fac = standin.velocity_fac(standin.cosmology(), 1.0)
    vel = box.realise_velocity(delta_k=fk)
    wv = bo.realise_velocity(geo, fk, fac)
    for c in range(3):
        assert np.max(np.abs(np.asarray(vel[c]) - wv[c])) < 10 * tol * np.std(wv[c])
    vel = box.realise_velocity(delta_x=fx)
    wv = bo.realise_velocity(geo, np.fft.fftn(fx), fac)
    assert np.max(np.abs(np.asarray(vel[2]) - wv[2])) < 10 * tol * np.std(wv[2])
    phi = np.asarray(box.realise_potential(delta_k=fk))
    assert np.max(np.abs(phi - bo.realise_potential(geo, fk))) < 10 * tol * np.std(phi)
    with pytest.raises(ValueError):
        box.realise_velocity(delta_x=fx, delta_k=fk)
    with pytest.raises(ValueError):
        box.binned_power_spectrum(delta_x=fx, delta_k=fk)


def test_interface_errors_and_limits():
    from fastbox_amd._lib import FastBoxError
    box = _box()
    with pytest.raises(FastBoxError):
        _box(N=34)                                   # a prime factor 17: refused, no fallback (2, 3, 5 only: test_generic_grid_gpu.py)
    with pytest.raises(FastBoxError):
        _box(N=8)
    with pytest.raises(ValueError):                  # scipy's griddata knows 'nearest', 'linear', 'cubic' for 1-D data
        box.redshift_space_density(delta_x=np.zeros((32,) * 3), velocity_z=np.zeros((32,) * 3), method="quintic")
    # method='cubic' with v = 0: the last sample wraps onto the first, two EQUAL abscissae -- scipy's spline raises
    # ("Expect x to not have duplicates"); the device marks such a line as not-a-number instead (fastbox_hip.h)
    flat = np.asarray(box.redshift_space_density(delta_x=np.ones((32,) * 3), velocity_z=np.zeros((32,) * 3), method="cubic"))
    assert not np.all(np.isfinite(flat))
    with pytest.raises(ValueError):
        box.binned_power_spectrum(delta_x=np.zeros((16,) * 3))      # wrong shape
    with pytest.raises(AttributeError):
        box.delta_k                                  # nothing realised yet, as in the reference
    with pytest.raises(FastBoxError):
        box.realise_density(); box.binned_power_spectrum(nbins=300)  # > 256 bins
    # unsorted edges: np.digitize raises in the reference; here the device rejects them
    with pytest.raises((FastBoxError, ValueError)):
        box.binned_power_spectrum(kbins=np.array([0.3, 0.1, 0.2]))


def test_empty_and_degenerate_bins():
    """Bins beyond the Nyquist sphere are empty -> NaN (box.py uses N, not N/2, for kmax);
    explicit edges starting at 0 keep the DC mode out of the fused log-normal path."""
    np.random.seed(1)
    box = _box(32, 5e2, precision="f64")
    dx = box.realise_density()
    kc, pk, err = box.binned_power_spectrum(nbins=20)
    assert np.isnan(pk[-1]) and np.isnan(pk[-2]) and not np.isnan(pk[5])
    geo = bo.box_geometry(5e2, 32)
    kb = np.array([0.0, 0.02, 0.05, 0.1, 0.2])
    ln = box.lognormal(dx)
    got = box.binned_power_spectrum(delta_x=ln, kbins=kb)
    want = bo.binned_power_spectrum(geo, np.fft.fftn(bo.lognormal(np.asarray(dx))), kbins=kb)
    assert np.allclose(got[1], want[1], rtol=1e-9, equal_nan=True)


@pytest.mark.parametrize("N", [512, 1024])
def test_full_size_properties(N):
    """Size-independent checks at the BASELINE sizes: Parseval, fused vs stand-alone binning,
    reproducibility of the device generator, log-normal >= -1 and its mean."""
    box = _box(N, 1e3, precision="f32", rng="device", seed=3)
    dx = box.realise_density()
    fused = box.binned_power_spectrum(delta_x=dx, nbins=20)        # fuses the z passes + binning
    s1, s2 = box.test_parseval()
    assert np.isclose(s1, s2, rtol=2e-5)
    plain = box.binned_power_spectrum(nbins=20)                    # stored spectrum, stand-alone kernel
    m = ~np.isnan(plain[1])
    assert np.array_equal(np.isnan(fused[1]), np.isnan(plain[1]))
    assert np.allclose(fused[1][m], plain[1][m], rtol=3e-6) and np.allclose(fused[2][m], plain[2][m], rtol=3e-5)
    mean = box.engine.sum_real(dx) / float(N) ** 3
    assert abs(mean) < 1e-3
    ln = box.lognormal(dx)
    total = box.engine.sum_real(ln) / float(N) ** 3
    assert abs(total) < 1e-4                                       # <exp(d)/mean - 1> = 0
    box2 = _box(N, 1e3, precision="f32", rng="device", seed=3)
    assert box2.engine.sum_real(box2.realise_density(), squared=True) == box.engine.sum_real(dx, squared=True)


def test_largest_size_properties():
    """2048^3 (34 GB of field, 36 GB of spectrum per buffer): the largest grid the plans support, on one GPU."""
    import gc
    N = 2048
    box = _box(N, 2e3, precision="f32", rng="device", seed=4)
    dx = box.realise_density()
    kc, pk, err = box.binned_power_spectrum(delta_x=dx, nbins=20)          # fused z passes + binning, plane batches
    m = ~np.isnan(pk)
    assert m.sum() >= 15 and np.all(pk[m] > 0) and np.all(err[m] >= 0)
    s1, s2 = box.test_parseval()
    assert np.isclose(s1, s2, rtol=2e-5)
    mean = box.engine.sum_real(dx) / float(N) ** 3
    assert abs(mean) < 1e-3
    # the estimate follows the input spectrum where shells are complete and well sampled (the last bins reach into
    # the corners of the cube beyond the Nyquist frequency); ~1 % low from averaging a falling P(k) over a wide bin
    kt, pt = box.theoretical_power_spectrum()
    ratio = pk[m][-9:-3] / np.interp(kc[m][-9:-3], kt, pt)
    assert np.all((ratio > 0.95) & (ratio < 1.02)), ratio
    sq = box.engine.sum_real(dx, squared=True)
    del dx, box
    gc.collect()
    box2 = _box(N, 2e3, precision="f32", rng="device", seed=4)
    assert box2.engine.sum_real(box2.realise_density(), squared=True) == sq          # bit-reproducible generator
    del box2
    gc.collect()


def test_many_queued_spectra_come_back_in_batches():
    """wait=False results live in a ring of device records fetched in one copy per batch; queue more than
    the ring holds (256) before asking for any, read them out of order, compare with one-at-a-time."""
    box = _box(32, 5e2, precision="f32", rng="device", seed=2)
    pend = [box.binned_power_spectrum(delta_x=box.realise_density(), nbins=12, wait=False) for _ in range(300)]
    order = list(range(299, -1, -7)) + list(range(300))
    got = {i: pend[i].result() for i in order}
    box2 = _box(32, 5e2, precision="f32", rng="device", seed=2)
    for i in range(300):
        want = box2.binned_power_spectrum(delta_x=box2.realise_density(), nbins=12)
        assert np.array_equal(got[i][1], want[1], equal_nan=True) and np.array_equal(got[i][2], want[2], equal_nan=True), i


def test_out_of_memory_releases_idle_buffers_and_retries():
    """hipMalloc failing is not final while engines hold pools of idle buffers (or dead boxes wait for a garbage
    collection): those are released and the allocation is tried once more."""
    import ctypes
    from fastbox_amd import _lib as lib_mod
    from fastbox_amd.device import _ENGINES
    box = _box(256, 1e3, precision="f32", rng="device", seed=1)
    eng = box.engine
    buf = eng._alloc_bytes(1 << 20)
    del buf                                              # -> idle in the pool
    assert eng._pool[1 << 20]
    real_call = lib_mod.call
    state = {"failed": False}

    def flaky(name, *args):                               # the first fb_malloc reports out of memory
        if name == "fb_malloc" and not state["failed"]:
            state["failed"] = True
            raise lib_mod.FastBoxError("fb_malloc", -4, "out of memory (injected)")
        return real_call(name, *args)

    lib_mod.call = flaky
    try:
        again = eng._alloc_bytes(3 << 20)
    finally:
        lib_mod.call = real_call
    assert state["failed"] and again.ptr and eng in _ENGINES
    assert not eng._pool[1 << 20]                         # the idle buffer went back to the driver
    dx = box.realise_density()                            # the engine is still good
    assert np.isfinite(eng.sum_real(dx, squared=True))


def test_stale_runtime_error_of_another_library_is_not_reported():
    """A failed HIP call elsewhere in the process leaves the runtime's per-thread "last error" set; the launch checks of
    this library must not mistake it for a failure of their own kernel."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    ptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(1 << 62)) != 0          # out of memory, left unread
    box = _box(64, 1e3, precision="f32", rng="device", seed=2)
    dx = box.realise_density()
    kc, pk, err = box.binned_power_spectrum(delta_x=dx, nbins=10)
    assert np.isfinite(pk[~np.isnan(pk)]).all()


def test_monte_carlo_checkpoint_resume_on_the_device(tmp_path):
    """fastbox_amd.montecarlo on a real box: a run killed after some batches resumes from its checkpoint and ends with
    the sums of the uninterrupted run, bit for bit (realisations are addressed by index: counter-based generator)."""
    from fastbox_amd import CosmoBox, default_cosmo, montecarlo
    mk = lambda: CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=64, realise_now=False, rng="device", seed=7000)
    full, kc, _ = montecarlo.run(mk(), 24, nbins=12, lognormal=True, batch=5)
    ck = str(tmp_path / "mc.npz")

    class Killed(Exception):
        pass

    def kill(done, total):
        if done >= 10:
            raise Killed()
    with pytest.raises(Killed):
        montecarlo.run(mk(), 24, nbins=12, lognormal=True, batch=5, checkpoint=ck, on_batch=kill)
    acc, kc2, _ = montecarlo.run(mk(), 24, nbins=12, lognormal=True, batch=5, checkpoint=ck)
    assert acc.n == 24 and np.array_equal(kc, kc2)
    assert np.array_equal(acc.mean, full.mean) and np.array_equal(acc.m2, full.m2)
    # two "ranks" on this GPU: the combined raw sums are those of one rank (to rounding of the merge)
    parts = [montecarlo.run(mk(), 24, nbins=12, lognormal=True, batch=5, rank=r, world=2)[0] for r in range(2)]
    tot = montecarlo.BandPowerAccumulator.from_raw_sums(*[sum(x) for x in zip(*[p.raw_sums() for p in parts])])
    assert np.allclose(tot.mean, full.mean, rtol=1e-12) and np.allclose(tot.covariance(), full.covariance(), rtol=1e-9, atol=0)


@pytest.mark.parametrize("N", [64, 512])
def test_spectrum_without_keeping_the_field(N):
    """binned_power_spectrum(..., keep_field=False) on a pending device-generator realisation: the fused z pass does not
    write delta_x; the spectrum is the same bit for bit, and reading the field afterwards draws the same realisation
    again (counter-based generator)."""
    from fastbox_amd import CosmoBox, default_cosmo
    mk = lambda: CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f32", rng="device", seed=21)
    a, b = mk(), mk()
    for ln in (False, True):
        da, db = a.realise_density(), b.realise_density()
        pa = a.binned_power_spectrum(delta_x=a.lognormal(da) if ln else da, nbins=20, wait=False)
        pb = b.binned_power_spectrum(delta_x=b.lognormal(db) if ln else db, nbins=20, wait=False, keep_field=False)
        assert da.materialised and not db.materialised
        for x, y in zip(pa.result(), pb.result()):
            assert np.array_equal(x, y, equal_nan=True)
        fa, fb = np.asarray(da), np.asarray(db)                  # the second one is regenerated here
        assert db.materialised and np.array_equal(fa, fb)


def test_one_box_waits_for_another_on_the_device():
    """Engine.wait_for (fb_stream_wait_stream): box B's stream reads a field that box A's stream is still
    producing -- ten queued realisations ahead of it -- without the host waiting in between."""
    from fastbox_amd import CosmoBox, default_cosmo
    from fastbox_amd.device import new_stream
    N = 256
    a = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f64", rng="device", seed=21,
                 stream=new_stream())
    b = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f64", rng="device", seed=22,
                 stream=new_stream())
    fields = [a.realise_density() for _ in range(10)]
    for f in fields:
        f.ptr                                          # queue every z pass on A's stream
    b.engine.wait_for(a.engine)
    got = b.engine.sum_real(fields[-1], squared=True)   # B's stream, A's buffer
    want = float(np.sum(np.asarray(fields[-1]) ** 2))
    assert abs(got - want) <= 1e-12 * want
