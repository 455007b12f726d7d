#!/usr/bin/env python
"""Redshift-space density field (cf. the reference's examples/example_redshift_space.py).  The reference's own lines
work unchanged (np.fft.ifftn(box.velocity_k[2]).real pulls the spectrum to the host); box.to_real() keeps the
velocity on the device.  python examples/example_redshift_space.py [nsamp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo


def main(nsamp=128, literal=False):
    np.random.seed(10)
    box = CosmoBox(cosmo=default_cosmo, box_scale=(1e2, 1e2, 1e2), nsamp=nsamp, realise_now=False)
    box.realise_density()
    box.realise_velocity()
    if literal:
        vel_z = np.fft.ifftn(box.velocity_k[2]).real          # exactly what the reference example writes
    else:
        vel_z = box.to_real(box.velocity_k[2])                # the same field, never leaves the GPU
    delta_s = box.redshift_space_density(delta_x=box.delta_x, velocity_z=vel_z, sigma_nl=200., method='linear')
    k, pk, _ = box.binned_power_spectrum(delta_x=box.delta_x)
    ks, pks, _ = box.binned_power_spectrum(delta_x=delta_s)
    good = ~np.isnan(pk)
    print("rms line-of-sight velocity %.1f km/s; P_s/P_r per bin:" % np.std(vel_z), np.round((pks / pk)[good][:8], 3))
    return np.asarray(delta_s)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 128)
