"""CPU: the Monte-Carlo driver's host logic (fastbox_amd/montecarlo.py) -- Welford sums against numpy, checkpoint /
resume ending in bit-identical sums, rank partition + raw-sum combination independent of the number of ranks."""
import os

import numpy as np
import pytest

from fastbox_amd import montecarlo


class _FakePending(object):
    def __init__(self, triple):
        self.triple = triple

    def result(self):
        return self.triple


class _FakeBox(object):
    """Stands in for a CosmoBox with rng='device': the spectrum of realisation r is a fixed function of r."""
    rng, seed, N, Lx, Ly, Lz = "device", 11, 16, 1e2, 1e2, 1e2

    def __init__(self):
        self._realisation = 0
        self.calls = []

    def realise_density(self):
        r = self._realisation
        self._realisation += 1
        self.calls.append(r)
        return r

    def lognormal(self, dx):
        return ("ln", dx)

    def binned_power_spectrum(self, delta_x, nbins, wait, keep_field=True):
        r = delta_x[1] if isinstance(delta_x, tuple) else delta_x
        g = np.random.RandomState(1000 + r)
        pk = 5. + g.normal(size=nbins - 1)
        pk[0] = np.nan                                  # empty first bin, as a real box gives
        return _FakePending((np.arange(nbins - 1) + 0.5, pk, np.zeros(nbins - 1)))


def _all_spectra(R, nbins=8):
    b = _FakeBox()
    return np.array([np.nan_to_num(b.binned_power_spectrum(r, nbins, False).result()[1]) for r in range(R)])


def test_welford_sums_match_numpy():
    acc, kc, dt = montecarlo.run(_FakeBox(), 37, nbins=8, batch=5)
    x = _all_spectra(37)
    assert acc.n == 37 and np.allclose(acc.mean, x.mean(axis=0), rtol=1e-13)
    assert np.allclose(acc.covariance(), np.cov(x.T), rtol=1e-11, atol=1e-14)


def test_checkpoint_resume_is_bit_identical(tmp_path):
    ck = str(tmp_path / "mc.npz")
    full, _, _ = montecarlo.run(_FakeBox(), 40, nbins=8, batch=6)

    class Killed(Exception):
        pass

    def kill_after(done, total):
        if done >= 18:
            raise Killed()
    box = _FakeBox()
    with pytest.raises(Killed):
        montecarlo.run(box, 40, nbins=8, batch=6, checkpoint=ck, on_batch=kill_after)
    assert os.path.exists(ck)
    box2 = _FakeBox()
    acc, kc, _ = montecarlo.run(box2, 40, nbins=8, batch=6, checkpoint=ck)
    assert box2.calls[0] == 18 and box2.calls[-1] == 39                       # continued, did not restart
    assert acc.n == 40 and np.array_equal(acc.mean, full.mean) and np.array_equal(acc.m2, full.m2)
    # a finished checkpoint makes a further call a no-op; a checkpoint of another run is refused
    again, _, _ = montecarlo.run(_FakeBox(), 40, nbins=8, batch=6, checkpoint=ck)
    assert again.n == 40 and np.array_equal(again.m2, full.m2)
    with pytest.raises(ValueError):
        montecarlo.run(_FakeBox(), 40, nbins=9, batch=6, checkpoint=ck)


def test_rank_partition_combines_to_the_single_rank_result():
    one, _, _ = montecarlo.run(_FakeBox(), 30, nbins=8, batch=4)
    parts = [montecarlo.run(_FakeBox(), 30, nbins=8, batch=4, rank=r, world=3)[0] for r in range(3)]
    n = sum(p.raw_sums()[0] for p in parts)
    s1 = sum(p.raw_sums()[1] for p in parts)
    s2 = sum(p.raw_sums()[2] for p in parts)
    tot = montecarlo.BandPowerAccumulator.from_raw_sums(n, s1, s2)
    assert tot.n == 30 and np.allclose(tot.mean, one.mean, rtol=1e-13)
    assert np.allclose(tot.covariance(), one.covariance(), rtol=1e-10, atol=1e-13)


def _combine_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        acc, kc, _ = montecarlo.run(_FakeBox(), 2, nbins=8, batch=4, rank=rank, world=world)    # rank 2 gets nothing
        assert (acc.n == 0) == (rank == 2)
        tot = montecarlo.combine(acc, dist)
        np.savez(os.path.join(out_dir, "cb%d.npz" % rank), n=tot.n, mean=tot.mean, m2=tot.m2)
    finally:
        dist.destroy_process_group()


def test_combine_with_a_rank_that_had_no_realisations(tmp_path):
    """world > realisations: the idle rank enters the all-reduce with zeros of the full length (it used to raise
    before the collective, leaving the other ranks waiting in it)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_combine_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    one, _, _ = montecarlo.run(_FakeBox(), 2, nbins=8, batch=4)
    for r in range(3):
        g = np.load(os.path.join(str(tmp_path), "cb%d.npz" % r))
        assert int(g["n"]) == 2 and np.allclose(g["mean"], one.mean, rtol=1e-13) and np.allclose(g["m2"], one.m2, atol=1e-12)
    empty = montecarlo.BandPowerAccumulator.from_raw_sums(0, np.zeros(3), np.zeros((3, 3)))
    assert empty.n == 0 and empty.mean is None


def test_checkpoint_of_another_spectrum_is_refused(tmp_path):
    ck = str(tmp_path / "mc.npz")

    class PBox(_FakeBox):
        redshift, scale_factor, amp = 0.0, 1.0, 1.0

        def _power(self, k, a, linear):
            return self.amp * k ** -1.5
    montecarlo.run(PBox(), 10, nbins=8, batch=5, checkpoint=ck)
    other = PBox()
    other.amp = 1.1
    with pytest.raises(ValueError):
        montecarlo.run(other, 10, nbins=8, batch=5, checkpoint=ck)
    acc, _, _ = montecarlo.run(PBox(), 10, nbins=8, batch=5, checkpoint=ck)          # the same physics resumes
    assert acc.n == 10
