"""The reference's end-to-end flow (examples/example_endtoend.py steps 1-4 and the high-pass of step 5) on one GPU,
everything resident in HBM: log-normal tracer field in redshift space -> brightness temperature cube -> + Gaussian
foregrounds -> + radiometer noise -> PCA cleaning -> k_par high-pass -> P(k).  The tracer bias / mean temperature (tracers.py) are
plain numbers here.  python tools/endtoend_bench.py [N]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, ForegroundModel, NoiseModel, BeamHighpass, filters

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
box = CosmoBox(cosmo=default_cosmo, box_scale=(4e3, 4e3, 4e3), nsamp=N, redshift=0.8, realise_now=False,
               precision="f32", rng="device", seed=10)
eng = box.engine
fg, noise_model = ForegroundModel(box), NoiseModel(box)
highpass = BeamHighpass(kpar0=0.009, kperp0=None, power=3.)          # example_endtoend.py:133
bias, Tb = 1.3, 0.12


def flow():
    dx = box.realise_density()                                                    # (1a)
    delta_ln = box.lognormal(dx * bias)                                           # (1b), (1c)
    vel_z = box.to_real(box.realise_velocity()[2])                                # (1d)
    delta_s = box.redshift_space_density(delta_x=delta_ln, velocity_z=vel_z, sigma_nl=120.)   # (1e)
    signal_cube = Tb * (1. + delta_s)                                             # (1f)
    fg_map = fg.realise_foreground_amp(amp=57., beta=1.1, monopole=10., smoothing_scale=4.)   # (2)
    alpha = fg.realise_spectral_index(mean_spec_idx=2.07, std_spec_idx=0.0002, smoothing_scale=15.)
    data_cube = signal_cube + fg.construct_cube(fg_map, alpha, freq_ref=130.)
    data_cube = data_cube + noise_model.realise_radiometer_noise(Tinst=18., tp=2., fov=1., Ndish=64)   # (3)
    data_cube = filters.pca_filter(data_cube, nmodes=4)                           # (4) PCA foreground cleaning
    cleaned = box.apply_transfer_fn(box.to_k(data_cube), highpass)                # (5) k_par high-pass
    return box.binned_power_spectrum(delta_x=cleaned.real, nbins=20, wait=False)


flow().result()
eng.sync(); t0 = time.perf_counter()
eng.profile_start()
out = [flow() for _ in range(5)]
res = [p.result() for p in out]
prof = eng.profile_stop()
dt = (time.perf_counter() - t0) / 5
print("N=%d: %.2f ms per end-to-end cube (%.1f /s)" % (N, dt * 1e3, 1 / dt))
print("   per-kernel-class ms:", {k: round(v[0] / 5, 3) for k, v in prof.items() if v[1]})
kc, pk, err = res[-1]
print("P(k) of the high-passed data cube:", np.array2string(pk[4:10], precision=4))
