"""CPU: the slab-decomposed driver (fastbox_amd.distributed.SlabBox) under gloo, world_size 2,
with numpy per-rank operations.  Checks the exchange logic (block layout, pack/unpack, k_y
offsets, all-reduce of the bin sums) against a single-process oracle computation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastbox_amd import hostgeom, rng
from oracle import box_oracle as bo
from oracle import standin

N, L, SEED = 16, 2e2, 5


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _expected():
    geo = bo.box_geometry(L, N)
    pk_fn = standin.pk_fn(standin.cosmology(), 1.0)
    z = rng.half_spectrum_noise(N, SEED, 0)
    k = bo.k_magnitude(geo)[:, :, :N // 2 + 1]
    amp = np.sqrt(np.nan_to_num(pk_fn(k.flatten())).reshape(k.shape) * geo["boxfactor"])
    dx = np.fft.irfftn(z * amp, s=(N, N, N), axes=(0, 1, 2))
    pk = bo.binned_power_spectrum(geo, np.fft.fftn(dx), nbins=12)
    pkln = bo.binned_power_spectrum(geo, np.fft.fftn(bo.lognormal(dx)), nbins=12)
    return dx, pk, pkln


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastbox_amd.distributed import SlabBox
        from tests.slab_numpy_ops import NumpySlabOps
        box = SlabBox(standin.DEFAULT_COSMO, box_scale=L, nsamp=N, seed=SEED,
                      ops_factory=lambda g, P, r: NumpySlabOps(g, P, r),
                      pk_fn=standin.pk_fn(standin.cosmology(), 1.0))
        dx = box.realise_density().numpy().copy()
        pk = box.binned_power_spectrum(nbins=12)
        pkln = box.binned_power_spectrum(nbins=12, lognormal=True)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), dx=dx, pk=np.array(pk), pkln=np.array(pkln))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_slab_driver_gloo(tmp_path, world):
    port = _free_port()
    if world == 1:
        _worker(0, 1, port, str(tmp_path))
    else:
        mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    want_dx, want_pk, want_pkln = _expected()
    got = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    dx = np.concatenate([g["dx"] for g in got], axis=0)
    assert dx.shape == (N, N, N)
    assert np.max(np.abs(dx - want_dx)) < 1e-12 * np.std(want_dx)
    for g in got:                                   # every rank holds the full, all-reduced P(k)
        for a, b in zip(g["pk"], np.array(want_pk)):
            assert np.allclose(a, b, rtol=1e-10, atol=0, equal_nan=True)
        for a, b in zip(g["pkln"][:2], np.array(want_pkln)[:2]):
            assert np.allclose(a, b, rtol=1e-10, atol=0, equal_nan=True)


def _mc_worker(rank, world, port, out_dir):
    """Monte-Carlo steps, synchronous and pipelined (wait=False: three realisations in flight, asynchronous
    all-to-alls overlapping the other realisations' passes), on the same seeds."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastbox_amd.distributed import SlabBox
        from tests.slab_numpy_ops import NumpySlabOps
        mk = lambda: SlabBox(standin.DEFAULT_COSMO, box_scale=L, nsamp=N, seed=SEED,
                             ops_factory=lambda g, P, r: NumpySlabOps(g, P, r),
                             pk_fn=standin.pk_fn(standin.cosmology(), 1.0))
        a, b = mk(), mk()
        sync = [a.realise_and_power(nbins=12, lognormal=(i % 2 == 1)) for i in range(7)]
        tickets = [b.realise_and_power(nbins=12, lognormal=(i % 2 == 1), wait=False) for i in range(7)]
        early = tickets[2].result()                  # resolving a ticket in the middle drains the pipeline
        more = [b.realise_and_power(nbins=12, wait=False) for _ in range(2)]
        b.flush()
        piped = [t.result() for t in tickets]
        assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(early, piped[2]))
        tail = [a.realise_and_power(nbins=12) for _ in range(2)]
        np.savez(os.path.join(out_dir, "mc%d.npz" % rank), sync=np.array(sync), piped=np.array(piped),
                 tail=np.array(tail), more=np.array([t.result() for t in more]),
                 dx_sync=a.delta_x.numpy().copy(), dx_piped=b.delta_x.numpy().copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])        # 8: the rank count north_star scores (every xGMI link of a GPU in use)
def test_pipelined_monte_carlo_equals_synchronous_steps(tmp_path, world):
    mp.spawn(_mc_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), "mc%d.npz" % r))
        assert np.array_equal(g["sync"], g["piped"], equal_nan=True)          # same kernels, same order of sums
        assert np.array_equal(g["tail"], g["more"], equal_nan=True)
        assert np.array_equal(g["dx_sync"], g["dx_piped"])                     # both hold the last realisation's slab
    g0 = np.load(os.path.join(str(tmp_path), "mc0.npz"))
    assert not np.array_equal(g0["sync"][0], g0["sync"][2], equal_nan=True)    # different realisations


def _interleave_worker(rank, world, port, out_dir):
    """Pipelined tickets with other calls of the box between them, and with `_realisation` set by the caller (as a
    resumed Monte-Carlo run does): no ticket may lose its buffers to a later call."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastbox_amd.distributed import SlabBox
        from tests.slab_numpy_ops import NumpySlabOps
        mk = lambda: SlabBox(standin.DEFAULT_COSMO, box_scale=L, nsamp=N, seed=SEED,
                             ops_factory=lambda g, P, r: NumpySlabOps(g, P, r),
                             pk_fn=standin.pk_fn(standin.cosmology(), 1.0))
        a, b = mk(), mk()
        order = [0, 3, 6, 1, 9, 12]                      # realisation indices: 0, 3, 6 and 9, 12 would all map to pair 0
        want = []
        for i in order:
            a._realisation = i
            want.append(a.realise_and_power(nbins=12))
        a._realisation = 40
        want_field = a.realise_density().numpy().copy()
        want_pk = a.binned_power_spectrum(nbins=12)
        tickets = []
        for n, i in enumerate(order):
            b._realisation = i
            tickets.append(b.realise_and_power(nbins=12, wait=False))
            if n == 2:                                   # between tickets that are still in flight
                b._realisation = 40
                got_field = b.realise_density().numpy().copy()
                got_pk = b.binned_power_spectrum(nbins=12)
        got = [t.result() for t in tickets]
        np.savez(os.path.join(out_dir, "il%d.npz" % rank), want=np.array(want), got=np.array(got),
                 want_field=want_field, got_field=got_field, want_pk=np.array(want_pk), got_pk=np.array(got_pk))
    finally:
        dist.destroy_process_group()


def test_tickets_survive_interleaved_calls_and_realisation_jumps(tmp_path):
    mp.spawn(_interleave_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        g = np.load(os.path.join(str(tmp_path), "il%d.npz" % r))
        assert np.array_equal(g["want"], g["got"], equal_nan=True)
        assert np.array_equal(g["want_field"], g["got_field"])
        assert np.array_equal(g["want_pk"], g["got_pk"], equal_nan=True)


def _repeat_worker(rank, world, port, out_dir):
    """A log-normal step whose exponentials leave the floating-point range (here: a shift that flushes every one of
    them to zero) is formed again with the shift from the field's maximum, on every rank, in all three call forms."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastbox_amd.distributed import SlabBox
        from tests.slab_numpy_ops import NumpySlabOps
        mk = lambda: SlabBox(standin.DEFAULT_COSMO, box_scale=L, nsamp=N, seed=SEED,
                             ops_factory=lambda g, P, r: NumpySlabOps(g, P, r),
                             pk_fn=standin.pk_fn(standin.cosmology(), 1.0))
        a, b = mk(), mk()
        want = [a.realise_and_power(nbins=12, lognormal=True) for _ in range(3)]
        a.realise_density()
        want.append(a.binned_power_spectrum(nbins=12, lognormal=True))
        assert a.ln_repeats == 0
        b._ln_shift = 2000.0                             # exp(d - 2000) = 0 in any precision
        b.ops.set_exp_shift(b._ln_shift)
        got = [b.realise_and_power(nbins=12, lognormal=True)]
        tickets = [b.realise_and_power(nbins=12, lognormal=True, wait=False) for _ in range(2)]
        got += [t.result() for t in tickets]
        b.realise_density()
        got.append(b.binned_power_spectrum(nbins=12, lognormal=True))
        assert b.ln_repeats == 4 and b.ops.exp_shift == 2000.0
        np.savez(os.path.join(out_dir, "rp%d.npz" % rank), want=np.array(want), got=np.array(got))
    finally:
        dist.destroy_process_group()


def test_out_of_range_lognormal_step_is_repeated_with_the_exact_shift(tmp_path):
    mp.spawn(_repeat_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        g = np.load(os.path.join(str(tmp_path), "rp%d.npz" % r))
        assert np.allclose(g["want"], g["got"], rtol=1e-9, atol=0, equal_nan=True)


def _chunk_worker(rank, world, port, out_dir):
    """One transform in C chunks along k_z (the all-to-all of a chunk runs beside the passes of the next): every
    call form, C = 1, 2, 4 (5 tiles: uneven chunks), against the unchunked box."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastbox_amd.distributed import SlabBox
        from tests.slab_numpy_ops import NumpySlabOps
        mk = lambda C: SlabBox(standin.DEFAULT_COSMO, box_scale=L, nsamp=N, seed=SEED, rank=rank, world=world,
                               ops_factory=lambda g, P, r: NumpySlabOps(g, P, r),
                               pk_fn=standin.pk_fn(standin.cosmology(), 1.0), chunks=C)
        out = {}
        for C in (1, 2, 4):
            b = mk(C)
            assert b.chunks == C
            dx0 = b.realise_density().numpy().copy()
            pk0 = b.binned_power_spectrum(nbins=12)
            ln0 = b.binned_power_spectrum(nbins=12, lognormal=True)
            rp = [b.realise_and_power(nbins=12, lognormal=(i == 1)) for i in range(2)]
            dx2 = b.delta_x.numpy().copy()
            d = b.realise_and_power(nbins=12, wait=False).result()
            out["dx0_%d" % C], out["dx2_%d" % C] = dx0, dx2
            out["pk_%d" % C] = np.array([pk0, ln0] + rp + [d])
        np.savez(os.path.join(out_dir, "ch%d_%d.npz" % (world, rank)), **out)
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_chunked_transform_equals_unchunked(tmp_path, world):
    if world == 1:
        _chunk_worker(0, 1, _free_port(), str(tmp_path))
    else:
        mp.spawn(_chunk_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "ch%d_%d.npz" % (world, r))) for r in range(world)]
    want_dx, want_pk, want_ln = _expected()
    full = np.concatenate([g["dx0_1"] for g in got], axis=0)
    assert np.max(np.abs(full - want_dx)) < 1e-12 * np.std(want_dx)                 # the unchunked box against the oracle
    for g in got:
        for C in (2, 4):
            assert np.array_equal(g["dx0_%d" % C], g["dx0_1"]) and np.array_equal(g["dx2_%d" % C], g["dx2_1"])   # fields: bit for bit
            assert np.allclose(g["pk_%d" % C], g["pk_1"], rtol=1e-12, atol=0, equal_nan=True)    # bin sums: another grouping
        for a, b in zip(g["pk_4"][0], np.array(want_pk)):
            assert np.allclose(a, b, rtol=1e-10, atol=0, equal_nan=True)
    if world > 1:                                                # the same numbers for every number of ranks
        one = tmp_path / "one"
        one.mkdir()
        _chunk_worker(0, 1, _free_port(), str(one))
        ref = np.load(str(one / "ch1_0.npz"))
        assert np.array_equal(np.concatenate([g["dx2_4"] for g in got], axis=0), ref["dx2_4"])
        assert np.allclose(got[0]["pk_4"], ref["pk_4"], rtol=1e-12, atol=0, equal_nan=True)


@pytest.mark.parametrize("world,C", [(2, 1), (4, 2), (8, 4)])
def test_virtual_ranks_in_one_process_equal_the_gloo_ranks(world, C):
    """VirtualComm (P ranks = P threads of one process; what the single-GPU tests at 2048^3 drive the shipped chunk
    sequencing with) gives the numbers of the real process group: field against the oracle, spectra on every rank."""
    from fastbox_amd.distributed import SlabBox, VirtualComm
    from tests.slab_numpy_ops import NumpySlabOps
    comm = VirtualComm(world)
    boxes = [SlabBox(standin.DEFAULT_COSMO, box_scale=L, nsamp=N, seed=SEED, rank=r, world=world, comm=comm, chunks=C,
                     ops_factory=lambda g, P, rr: NumpySlabOps(g, P, rr), pk_fn=standin.pk_fn(standin.cosmology(), 1.0))
             for r in range(world)]
    want_dx, want_pk, want_ln = _expected()
    dx = comm.run(boxes, lambda b: b.realise_density().numpy().copy())
    assert np.max(np.abs(np.concatenate(dx, axis=0) - want_dx)) < 1e-12 * np.std(want_dx)
    pk = comm.run(boxes, lambda b: b.binned_power_spectrum(nbins=12))
    ln = comm.run(boxes, lambda b: b.binned_power_spectrum(nbins=12, lognormal=True))
    for r in range(world):
        for a, b in zip(pk[r], want_pk):
            assert np.allclose(a, b, rtol=1e-10, atol=0, equal_nan=True)
        for a, b in zip(ln[r], want_ln):
            assert np.allclose(a, b, rtol=1e-9, atol=0, equal_nan=True)
    # a failing rank releases the others instead of leaving them in the rendezvous
    def boom(b):
        if b.rank == 1:
            raise RuntimeError("rank 1 fails")
        return b.realise_density()
    with pytest.raises((RuntimeError, Exception)):
        comm.run(boxes, boom)


def test_shell_thresholds_reproduce_digitize():
    """Host tables handed to the device: the threshold form equals np.digitize on every shell
    that is not flagged ambiguous, for several box sizes (edge-on-a-shell cases included)."""
    for Lside in (1e2, 1e3, 2e3, 4e3):
        g = hostgeom.grid(Lside, 64)
        for nb in (20, 50):
            bins, kc = hostgeom.bin_edges(g, nb)
            thr, amb = hostgeom.shell_thresholds(64, g["L"][0], bins)
            k = hostgeom.shell_wavenumbers(64, g["L"][0])
            want = np.digitize(k, bins)
            got = np.searchsorted(thr, np.arange(k.size), side="right")
            ok = np.ones(k.size, bool); ok[list(amb)] = False
            assert np.array_equal(got[ok], want[ok])
