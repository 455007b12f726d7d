"""GPU sanity sweep: every public path at every supported grid size and both precisions (finite results, fused
paths against their step-by-step forms).  Catches launch-configuration limits (LDS, registers) that only bite at
particular sizes.  python tools/matrix_check.py [max_N]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, Wedge, ForegroundModel, NoiseModel, filters

maxN = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bad = 0
for N in (16, 32, 64, 128, 256, 512, 1024, 2048):
    if N > maxN:
        break
    for prec in ("f32", "f64"):
        if N >= 2048 and prec == "f64":
            continue
        t0 = time.time()
        tol = 2e-5 if prec == "f32" else 1e-10
        try:
            box = CosmoBox(default_cosmo, box_scale=1e3, nsamp=N, redshift=0.5, realise_now=False, precision=prec, rng="device", seed=N)
            eng = box.engine
            dx = box.realise_density()
            kc, pk, err = box.binned_power_spectrum(delta_x=box.lognormal(dx), nbins=20)          # fused chain
            kc2, pk2, err2 = box.binned_power_spectrum(delta_x=box.lognormal(dx), nbins=20)       # dx materialised now
            m = ~np.isnan(pk2)
            assert np.allclose(pk[m], pk2[m], rtol=20 * tol), "fused vs materialised P(k)"
            s1, s2 = box.test_parseval()
            assert abs(s1 / s2 - 1) < 50 * tol, "Parseval"
            vz = box.to_real(box.realise_velocity()[2])                                           # regenerated
            if N < 2048:       # the stored-spectrum route holds three more 34-36 GB cubes: beyond 288 GB with the rest
                vz2 = box.to_real(box.realise_velocity(delta_x=dx, inplace=False)[2])
                assert eng.sum_real(vz - vz2, squared=True) <= (40 * tol) ** 2 * eng.sum_real(vz2, squared=True), "velocity"
            vz2 = None                                                  # (2048^3: a cube is 34 GB, keep few alive)
            ds = box.redshift_space_density(delta_x=dx, velocity_z=vz, sigma_nl=100.)
            vz = None
            lazy = box.apply_transfer_fn(box.to_k(ds), Wedge(0.3))
            kcf, pkf, _ = box.binned_power_spectrum(delta_x=lazy.real, nbins=20)                  # BINF
            filt = lazy.real
            kcg, pkg, _ = box.binned_power_spectrum(delta_x=eng.upload(np.asarray(filt), "real") if N <= 256 else filt + 0., nbins=20)
            mf = ~np.isnan(pkg)
            assert np.allclose(pkf[mf], pkg[mf], rtol=50 * tol, atol=1e-30), "filtered P(k)"
            ds = lazy = filt = None
            if N < 2048:       # (the steps after the path at 2048^3 want a 68 GB complex cube on top of everything else)
                fg = ForegroundModel(box)
                cube = fg.construct_cube(fg.realise_foreground_amp(57., 1.1, 10., 4.), fg.realise_spectral_index(2.07, 2e-4, 15.)) \
                    + NoiseModel(box).realise_radiometer_noise(18., 2., 1., 64) + 0.1 * dx
                clean = filters.pca_filter(cube, 2)
                cube = None
                assert np.isfinite(eng.sum_real(clean, squared=True)), "pca"
                bp = filters.angular_bandpass_filter(clean, 0.05, 0.3)
                assert np.isfinite(float(eng.sumsq_half(eng.crop_full(bp))) if hasattr(eng, "sumsq_half") else 0.0)
            print("N=%4d %s ok  (%.1f s)" % (N, prec, time.time() - t0))
        except Exception as e:                                   # keep sweeping: report everything that breaks
            bad += 1
            print("N=%4d %s FAILED: %s: %s" % (N, prec, type(e).__name__, str(e)[:200]))
        finally:
            box = eng = dx = vz = vz2 = ds = lazy = filt = cube = clean = bp = None
print("failures:", bad)
sys.exit(1 if bad else 0)
