"""Tuning aid (GPU): BASELINE config 3 with 1, 2 or 3 boxes per GPU (independent chains on their own streams) and different
plane batches, all in one process and interleaved, so that the comparison does not depend on the box of the pool or its clocks."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, Wedge
from fastbox_amd.device import new_stream
N = 512
def mk(n, pb):
    bs = [CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f32", rng="device", seed=5 + i,
                   stream=(new_stream(0, None) if n > 1 else None)) for i in range(n)]
    if pb is not None:
        for b in bs: b.engine.set_plane_batching(pb, 1)
    return bs
wedge = Wedge(slope=0.3)
def chain(box):
    dx = box.realise_density()
    vz = box.to_real(box.realise_velocity()[2])
    ds = box.redshift_space_density(delta_x=dx, velocity_z=vz, sigma_nl=0.)
    filt = box.apply_transfer_fn(box.to_k(ds), wedge)
    pk = box.binned_power_spectrum(delta_x=filt.real, nbins=20, wait=False)
    filt.ptr
    return pk
cfgs = {"1 box, whole-box passes": mk(1, None), "2 boxes, 66-plane batches": mk(2, 66), "2 boxes, 32-plane batches": mk(2, 32),
        "2 boxes, whole-box passes": mk(2, 0), "3 boxes, 44-plane batches": mk(3, 44)}
res = {k: [] for k in cfgs}
for rnd in range(4):
    for name, bs in cfgs.items():
        for p in [chain(bs[i % len(bs)]) for i in range(6)]: p.result()
        for b in bs: b.engine.sync()
        t0 = time.perf_counter()
        out = [p for p in [chain(bs[i % len(bs)]) for i in range(30)]]
        for p in out: p.result()
        for b in bs: b.engine.sync()
        if rnd: res[name].append((time.perf_counter() - t0) / 30 * 1e3)
for k, v in res.items():
    print("%-28s ms per chain: %s" % (k, " ".join("%.3f" % x for x in v)))
