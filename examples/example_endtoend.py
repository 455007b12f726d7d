#!/usr/bin/env python
"""The reference's end-to-end flow (examples/example_endtoend.py) on the GPU: log-normal tracer field in redshift
space -> brightness temperature -> + foregrounds -> + radiometer noise -> PCA cleaning -> k_par high-pass -> P(k).
The tracer bias and mean temperature (fastbox.tracers.HITracer) are plain numbers here.
python examples/example_endtoend.py [nsamp]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, ForegroundModel, NoiseModel, BeamHighpass, filters


def main(nsamp=128):
    t0 = time.time()
    np.random.seed(10)
    box = CosmoBox(cosmo=default_cosmo, box_scale=(4e3, 4e3, 4e3), nsamp=nsamp, redshift=0.8, realise_now=False)
    box.realise_density()                                                             # (1a)
    bias, Tb = 1.3, 0.12
    delta_ln = box.lognormal(box.delta_x * bias)                                      # (1b), (1c)
    vel_z = box.to_real(box.realise_velocity(delta_x=box.delta_x, inplace=True)[2])   # (1d)
    delta_s = box.redshift_space_density(delta_x=delta_ln, velocity_z=vel_z, sigma_nl=120., method='linear')   # (1e)
    signal_cube = Tb * (1. + delta_s)                                                 # (1f)
    fg = ForegroundModel(box)                                                         # (2)
    fg_map = fg.realise_foreground_amp(amp=57., beta=1.1, monopole=10., smoothing_scale=4., redshift=box.redshift)
    alpha = fg.realise_spectral_index(mean_spec_idx=2.07, std_spec_idx=0.0002, smoothing_scale=15., redshift=box.redshift)
    data_cube = signal_cube + fg.construct_cube(fg_map, alpha, freq_ref=130., redshift=box.redshift)
    data_cube += NoiseModel(box).realise_radiometer_noise(Tinst=18., tp=2., fov=1., Ndish=64)     # (3)
    cleaned4, U_fg, amp_fg = filters.pca_filter(data_cube, nmodes=4, return_filter=True)          # (4)
    highpass = BeamHighpass(kpar0=0.009, kperp0=None, power=3.)                       # (5): 1 - exp(-0.5 (|k_par|/0.009)^3)
    cleaned4_hp = box.apply_transfer_fn(box.to_k(cleaned4), transfer_fn=highpass)
    k, pk_true, _ = box.binned_power_spectrum(delta_x=signal_cube - Tb)
    k, pk_clean, _ = box.binned_power_spectrum(delta_x=cleaned4_hp.real)
    good = ~np.isnan(pk_true)
    print("%d^3 end-to-end in %.2f s; P_cleaned / P_signal per bin:" % (nsamp, time.time() - t0),
          np.round((pk_clean / pk_true)[good][3:11], 3))
    return np.asarray(cleaned4)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 128)
