"""
Monte-Carlo band-power covariance (BASELINE.json configs[4]: R realisations of an N^3 box, P(k) of each, mean and
covariance of the band powers) with checkpoint / resume.

The reference has no such driver (and no checkpointing: SURVEY.md 5); its users write the loop
``for r in range(R): box.realise_density(); box.binned_power_spectrum()`` by hand.  Here the loop state is tiny and
exact: the device generator is counter-based, so a run is fully described by (seed, next realisation index) plus the
Welford accumulators (n, mean, M2) of the band powers -- a few KB.  A killed 1000 x 2048^3 run resumes at the next
realisation and ends with bit-identical sums, because every realisation is addressed by its index, not by stream
position.

One process per GPU; ranks take realisations r with r % world == rank and combine (n, mean, M2) at the end (Chan et
al. pairwise update through raw sums): no data-path collective.
"""
import json
import os
import time

import numpy as np


class BandPowerAccumulator(object):
    """Welford mean / co-moment sums of band-power vectors, fp64 on the host."""

    def __init__(self, size=None):
        self.n, self.mean, self.m2 = 0, None, None
        self.size = size            # length of the vectors to come (known to `run` before the first one arrives)

    def add(self, x):
        x = np.nan_to_num(np.asarray(x, dtype=np.float64))
        if self.mean is None:
            self.mean, self.m2 = np.zeros_like(x), np.zeros((x.size, x.size))
        self.n += 1
        d = x - self.mean
        self.mean += d / self.n
        self.m2 += np.outer(d, x - self.mean)

    def covariance(self):
        return self.m2 / (self.n - 1)

    def raw_sums(self):
        """(n, n mean, M2 + n mean mean^T): additive over disjoint sets of realisations."""
        return self.n, self.n * self.mean, self.m2 + self.n * np.outer(self.mean, self.mean)

    @classmethod
    def from_raw_sums(cls, n, s1, s2):
        a = cls()
        a.n = int(n)
        if a.n == 0:
            return a
        a.mean = s1 / n
        a.m2 = s2 - n * np.outer(a.mean, a.mean)
        return a


def _save(path, state, acc, kc):
    tmp = path + ".tmp"
    np.savez(tmp, meta=json.dumps(state), n=acc.n, mean=acc.mean if acc.mean is not None else np.zeros(0),
             m2=acc.m2 if acc.m2 is not None else np.zeros((0, 0)), kc=kc if kc is not None else np.zeros(0))
    os.replace(tmp + ".npz" if os.path.exists(tmp + ".npz") else tmp, path)       # atomic: a kill never leaves half a file


def _load(path):
    g = np.load(path, allow_pickle=False)
    acc = BandPowerAccumulator()
    acc.n = int(g["n"])
    if acc.n:
        acc.mean, acc.m2 = g["mean"].copy(), g["m2"].copy()
    return json.loads(str(g["meta"])), acc, (g["kc"].copy() if g["kc"].size else None)


def _physics_id(box):
    """What else decides the numbers of a run: the input spectrum (through the amplitudes the box would draw with, at
    its own redshift and its default non-linear P(k)), the plan's precision.  A checkpoint written with another
    cosmology, redshift or precision must not be continued."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((getattr(box, "redshift", None), getattr(getattr(box, "engine", None), "precision", None))).encode())
    power = getattr(box, "_power", None)
    if power is not None:
        k = 2. * np.pi * np.sqrt(np.arange(1, 3 * (int(box.N) // 2) ** 2 + 1, 97, dtype=np.float64)) / float(box.Lx)
        with np.errstate(all="ignore"):
            h.update(np.nan_to_num(np.asarray(power(k, getattr(box, "scale_factor", 1.0), False), dtype=np.float64)).tobytes())
    return h.hexdigest()[:16]


def run(box, realisations, nbins=20, lognormal=False, batch=50, rank=0, world=1, checkpoint=None,
        checkpoint_every=1, on_batch=None, keep_fields=False):
    """P(k) of realisations r = rank, rank + world, ... < `realisations` of ``box`` (a CosmoBox with rng='device').

    ``checkpoint``: file of this rank's state, written after every ``checkpoint_every`` batches and read at the start
    if it exists and matches (seed, grid, bins, lognormal, rank, world): the loop continues at the recorded
    realisation.  Returns (accumulator, k centres, seconds spent in this call)."""
    if getattr(box, "rng", "device") != "device":
        raise ValueError("the Monte-Carlo driver needs rng='device' (realisations addressed by index)")
    ident = dict(seed=int(box.seed), nsamp=int(box.N), nbins=int(nbins), lognormal=bool(lognormal), rank=int(rank),
                 world=int(world), box=[float(box.Lx), float(box.Ly), float(box.Lz)], physics=_physics_id(box))
    acc, kc, done = BandPowerAccumulator(int(nbins) - 1), None, 0
    if checkpoint and os.path.exists(checkpoint):
        meta, acc0, kc0 = _load(checkpoint)
        if {k: meta.get(k) for k in ident} != ident:
            raise ValueError("checkpoint %s belongs to another run: %s" % (checkpoint, meta))
        acc, kc, done = acc0, kc0, int(meta["done"])
        acc.size = int(nbins) - 1
    mine = [r for r in range(realisations) if r % world == rank]
    # the covariance needs the spectra only: the fused z pass does not write delta_x (it can be drawn again by index)
    spectra_only = {} if keep_fields else {"keep_field": False}
    t0 = time.perf_counter()
    nb = 0
    for start in range(done, len(mine), batch):
        chunk = mine[start:start + batch]
        if not keep_fields and hasattr(box, "realisation_spectra"):
            # the whole batch queued by one library call (fb_montecarlo_power): the same numbers as the loop below
            box._realisation = chunk[0]
            kc, pks, _ = box.realisation_spectra(len(chunk), nbins=nbins, lognormal=lognormal, stride=world)
            for pk in pks:
                acc.add(pk)
        else:
            pend = []
            for r in chunk:
                box._realisation = r                     # the generator's counter: realisation r, whatever ran before
                dx = box.realise_density()
                pend.append(box.binned_power_spectrum(delta_x=box.lognormal(dx) if lognormal else dx, nbins=nbins, wait=False,
                                                      **spectra_only))
            for p in pend:
                kc, pk, _ = p.result()
                acc.add(pk)
        nb += 1
        if checkpoint and (nb % checkpoint_every == 0 or start + batch >= len(mine)):
            _save(checkpoint, dict(ident, done=start + len(chunk)), acc, kc)
        if on_batch is not None:
            on_batch(start + len(chunk), len(mine))
    return acc, kc, time.perf_counter() - t0


def combine(acc, dist=None, device=None, size=None):
    """All-reduce the raw sums over the ranks of torch.distributed (every rank gets the combined accumulator).
    A rank that processed no realisation (more ranks than realisations) enters the collective with zeros of the full
    length, which it takes from the accumulator `run` returned (``acc.size``) or from ``size`` (= nbins - 1)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return acc
    import torch
    if acc.mean is not None:
        if size is not None and int(size) != acc.mean.size:
            raise ValueError("size=%d but this rank's band-power vectors have %d entries" % (size, acc.mean.size))
        size = acc.mean.size
    if size is None:
        size = getattr(acc, "size", None)
    if size is None:
        raise ValueError("combine(): this rank has no realisations; pass size=nbins - 1")
    size = int(size)
    if acc.n:
        n, s1, s2 = acc.raw_sums()
    else:
        n, s1, s2 = 0, np.zeros(size), np.zeros((size, size))
    t = torch.tensor(np.concatenate([[n], np.ravel(s1), np.ravel(s2)]), dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t)
    t = t.cpu().numpy()
    return BandPowerAccumulator.from_raw_sums(t[0], t[1:1 + size], t[1 + size:].reshape(size, size))
