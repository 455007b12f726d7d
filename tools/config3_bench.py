"""BASELINE config 3 on one GPU: 512^3 box, gen -> v_z -> redshift-space remap -> k_perp/k_par
foreground-wedge filter -> P(k) of the filtered field, everything resident in HBM.  Prints the time per
kernel class (HIP events; one box) and per whole chain.

    python tools/config3_bench.py [N] [boxes] [plane batch] [plane streams]

boxes > 1: independent chains round-robin on that many boxes, each on its own HIP stream (as bench.py's headline)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, Wedge
from fastbox_amd.device import new_stream

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 1
PB = int(sys.argv[3]) if len(sys.argv) > 3 else None
PS = int(sys.argv[4]) if len(sys.argv) > 4 else 0
boxes = [CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f32", rng="device", seed=5 + i,
                  stream=(new_stream(0, 0) if NB > 1 else None)) for i in range(NB)]
if PB is not None:
    for b in boxes:
        b.engine.set_plane_batching(PB, PS)
wedge = Wedge(slope=0.3)
count = [0]

def chain(sigma_nl):
    box = boxes[count[0] % NB]
    count[0] += 1
    dx = box.realise_density()
    vz = box.to_real(box.realise_velocity()[2])
    ds = box.redshift_space_density(delta_x=dx, velocity_z=vz, sigma_nl=sigma_nl)
    dk = box.to_k(ds)                                       # pending forward transform
    filt = box.apply_transfer_fn(dk, wedge)                 # lazy (Hermitian field, filter even in k_par)
    pk = box.binned_power_spectrum(delta_x=filt.real, nbins=20, wait=False)
    filt.ptr                                                # deliver the filtered field as well
    return pk

REP = 20
for sigma in (0.0, 200.0):
    for p in [chain(sigma) for _ in range(2 * NB)]:
        p.result()
    for b in boxes:
        b.engine.sync()
    t0 = time.perf_counter()
    if NB == 1:
        boxes[0].engine.profile_start()
    pend = [chain(sigma) for _ in range(REP)]
    out = [p.result() for p in pend]
    for b in boxes:
        b.engine.sync()
    dt = (time.perf_counter() - t0) / REP
    print("sigma_nl=%5.1f: %.3f ms per chain (%.1f chains/s), %d box(es)" % (sigma, dt * 1e3, 1 / dt, NB))
    if NB == 1:
        prof = boxes[0].engine.profile_stop()
        print("   per-kernel-class ms per chain:", {k: round(v[0] / REP, 3) for k, v in prof.items() if v[1]})
kc, pk, err = out[-1]
assert np.all(np.isfinite(pk[~np.isnan(pk)]))
print("P(k) of the filtered redshift-space field:", np.round(pk[3:9], 2))
