"""One-off evidence (GPU box, ~2 min of host time, ~25 GB of host memory): parity of the device path with the oracle at
the BASELINE size, same numpy seed.  python tools/parity_fullsize.py [N]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox
from oracle import box_oracle as bo, standin

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
seed, L = 14, 1e3
t0 = time.time()
geo = bo.box_geometry(L, N)
re, im = bo.draw_noise(N, np.random.RandomState(seed))
odx, odk = bo.realise_density(geo, standin.pk_fn(standin.cosmology(), 1.0), re, im)
okc, opk, oerr = bo.binned_power_spectrum(geo, odk)
oln = bo.lognormal(odx)
okl, opl, oel = bo.binned_power_spectrum(geo, np.fft.fftn(oln))
del re, im, odk
print("oracle (numpy, 1 thread) %d^3: %.1f s" % (N, time.time() - t0))
m = ~np.isnan(opk)
for prec in ("f32", "f64"):
    np.random.seed(seed)
    box = CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=L, nsamp=N, realise_now=False, precision=prec)
    t1 = time.time()
    dx = box.realise_density()
    kc, pk, err = box.binned_power_spectrum()
    kl, pl, el = box.binned_power_spectrum(delta_x=box.lognormal(dx))
    t2 = time.time()
    hdx = np.asarray(dx)
    print("%s: delta_x max|diff|/sigma = %.2e   P(k) max rel = %.2e   stddev max rel = %.2e   log-normal P(k) max rel = %.2e   "
          "NaN masks equal: %s   (device path incl. host normals %.1f s)"
          % (prec, np.max(np.abs(hdx - odx)) / np.std(odx), np.max(np.abs(pk[m] / opk[m] - 1)),
             np.max(np.abs(err[m] / oerr[m] - 1)), np.max(np.abs(pl[m] / opl[m] - 1)),
             np.array_equal(np.isnan(pk), np.isnan(opk)) and np.array_equal(kc, okc), t2 - t1))
    del box, dx, hdx
