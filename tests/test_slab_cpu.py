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


@pytest.mark.parametrize("world", [2, 4])
def test_pipelined_monte_carlo_equals_synchronous_steps(tmp_path, world):
    mp.spawn(_mc_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), "mc%d.npz" % r))
        assert np.array_equal(g["sync"], g["piped"], equal_nan=True)          # same kernels, same order of sums
        assert np.array_equal(g["tail"], g["more"], equal_nan=True)
        assert np.array_equal(g["dx_sync"], g["dx_piped"])                     # both hold the last realisation's slab
    g0 = np.load(os.path.join(str(tmp_path), "mc0.npz"))
    assert not np.array_equal(g0["sync"][0], g0["sync"][2], equal_nan=True)    # different realisations


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
