"""GPU: the two schedules of the strided FFT passes (fb_set_pass_schedule: one workgroup per tile, or resident
workgroups that walk the tiles and load their next tile while finishing the current one) are the same arithmetic:
fields bit-identical, bin sums equal to fp64 rounding of a different grouping of the per-workgroup partial sums."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(N, precision, schedule, seed=17, L=1e3):
    from fastbox_amd import CosmoBox, default_cosmo
    box = CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, precision=precision, rng="device", seed=seed)
    box.engine.set_pass_schedule(*schedule)
    dx = box.realise_density()
    ln = box.binned_power_spectrum(delta_x=box.lognormal(dx), nbins=20)            # GEN, y, fused z, y, BIN
    field = np.asarray(dx).astype(np.float32 if precision == "f32" else np.float64)
    ga = box.binned_power_spectrum(delta_x=dx, nbins=20)                             # r2c, y, BIN from a stored field
    dk = box.engine.download_half_raw(box.delta_k)                                    # r2c, y, x (plain, stored)
    back = np.asarray(box.engine.fft_c2r(box.delta_k))                                # x, y, c2r (plain)
    box.engine.close()
    return field, ln, ga, dk, back


@pytest.mark.parametrize("N,precision", [(256, "f32"), (512, "f32"), (256, "f64"), (1024, "f32")])
def test_resident_schedule_equals_one_tile_per_workgroup(N, precision):
    if N == 1024:
        want = _run(N, precision, (0, 0, 0))
        got = _run(N, precision, (1, 1, 1))
        cases = [((1, 1, 1), got)]
    else:
        want = _run(N, precision, (0, 0, 0))
        cases = [(s, _run(N, precision, s)) for s in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1))]
    for sched, got in cases:
        assert np.array_equal(got[0], want[0]), sched                                 # delta_x, bit for bit
        assert np.array_equal(got[3], want[3]) and np.array_equal(got[4], want[4]), sched
        for a, b in ((got[1], want[1]), (got[2], want[2])):
            assert np.array_equal(a[0], b[0]) and np.array_equal(np.isnan(a[1]), np.isnan(b[1]))
            m = ~np.isnan(b[1])
            assert np.allclose(a[1][m], b[1][m], rtol=1e-12, atol=0), sched
            assert np.allclose(a[2][m], b[2][m], rtol=1e-9, atol=1e-12 * b[1][m].max()), sched


def test_generator_with_parked_stores_at_2048():
    """N = 2048 (16 points per thread): the resident generator pass keeps its finished tile in registers and stores it two
    rows at a time between the next tile's batches and stages.  Against one workgroup per tile, which stores at once:
    sum, sum of squares and maximum of delta_x (fixed-order device reductions) equal bit for bit, P(k) to fp64 rounding."""
    from fastbox_amd import CosmoBox, default_cosmo
    out = []
    for sched in ((0, 0, 0), (-1, -1, -1)):
        box = CosmoBox(cosmo=default_cosmo, box_scale=2e3, nsamp=2048, realise_now=False, precision="f32", rng="device", seed=23)
        box.engine.set_pass_schedule(*sched)
        dx = box.realise_density()
        pk = box.binned_power_spectrum(delta_x=dx, nbins=20)
        eng = box.engine
        out.append((eng.sum_real(dx), eng.sum_real(dx, squared=True), eng.max_real(dx), pk))
        del dx
        eng.close()
    a, b = out
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
    m = ~np.isnan(a[3][1])
    assert np.array_equal(a[3][0], b[3][0]) and np.allclose(a[3][1][m], b[3][1][m], rtol=1e-12, atol=0)


@pytest.mark.parametrize("P", [1, 2, 8])
def test_resident_schedule_in_the_slab_path(P):
    """The exchange-buffer addressing of the slab-decomposed transform (a line cut into per-rank pieces) under the
    resident schedule: virtual ranks on one GPU, fields and bin sums against the one-tile schedule."""
    from fastbox_amd import default_cosmo
    from fastbox_amd.distributed import HipSlabOps, SlabBox, run_virtual
    N, L, seed, nb = 256, 1e3, 5, 20
    out = []
    for sched in ((0, 0, 0), (1, 1, 1)):
        boxes = [SlabBox(default_cosmo, box_scale=L, nsamp=N, precision="f32", seed=seed, rank=r, world=P,
                         ops_factory=lambda g, PP, rr: HipSlabOps(g, PP, rr, precision="f32", device=0)) for r in range(P)]
        for b in boxes:
            b.ops.engine.set_pass_schedule(*sched)
            b._pk_setup(nb, None)

        def turn(b, recv):
            b._res = b.ops.new_results(2 * nb + 1)
            b.delta_x = b.ops.new_real()
            b._send2 = b._kslab if recv is b._xbuf else b._xbuf
            b.ops.turnaround(recv, b._half, b.delta_x, b._send2, True, b._res[2 * nb:])
            return b._send2
        run_virtual(boxes, lambda b: b._gen_local(), turn)
        dx = np.concatenate([b.delta_x.cpu().numpy() for b in boxes], axis=0)
        res = run_virtual(boxes, lambda b: b._send2, lambda b, kslab: b._pk_finish(kslab, nb).clone())
        out.append((dx, sum(r.cpu().numpy() for r in res)))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.allclose(out[0][1], out[1][1], rtol=1e-12, atol=0)


def test_wide_row_form_equals_narrow_form_at_2048():
    """N = 2048, single precision: the strided passes in 128-byte rows (2048 x 16 tile exchanged as real / imaginary halves, 32
    points per thread -- the default) against the 64-byte-row form of rounds 1-3 (fb_set_tile_rows(64)): the same radix-8/8/8/4
    stages, so delta_x agrees bit for bit (sum, sum of squares, maximum by fixed-order device reductions; a probe plane compared
    element by element); both spectra -- Gaussian through the r2c route, log-normal through the fused z pass -- to 1e-7: a lane of
    the binning pass sums its 16 or 32 modes in single precision before the fp64 wave rows, so the two forms group differently
    (measured 1e-9).  Both schedules of the new form."""
    from fastbox_amd import CosmoBox, default_cosmo
    out = []
    for rows, sched in ((64, (-1, -1, -1)), (128, (-1, -1, -1)), (128, (0, 0, 0))):
        box = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=2048, realise_now=False, precision="f32", rng="device", seed=29)
        box.engine.set_tile_rows(rows)
        box.engine.set_pass_schedule(*sched)
        dx = box.realise_density()
        ln = box.binned_power_spectrum(delta_x=box.lognormal(dx), nbins=20)           # GEN, y, fused z, y, BIN
        pk = box.binned_power_spectrum(delta_x=dx, nbins=20)                            # r2c, y, BIN
        eng = box.engine
        probe = np.stack([eng.download_plane(dx, ix) for ix in (0, 1000, 2047)])
        out.append((eng.sum_real(dx), eng.sum_real(dx, squared=True), eng.max_real(dx), probe, pk, ln))
        del dx
        eng.close()
    a = out[0]
    for b in out[1:]:
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
        assert np.array_equal(a[3], b[3])
        for q in (4, 5):
            m = ~np.isnan(a[q][1])
            assert np.array_equal(a[q][0], b[q][0]) and np.array_equal(np.isnan(a[q][1]), np.isnan(b[q][1]))
            assert np.allclose(a[q][1][m], b[q][1][m], rtol=1e-7, atol=0), np.max(np.abs(a[q][1][m] / b[q][1][m] - 1))
            assert np.allclose(a[q][2][m], b[q][2][m], rtol=1e-6, atol=1e-9 * a[q][1][m].max())
