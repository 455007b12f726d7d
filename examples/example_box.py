#!/usr/bin/env python
"""A Gaussian box, its log-normal transform and their power spectra (cf. the reference's examples/example_box.py):
the same calls as with fastbox.box, running on the GPU.  python examples/example_box.py [nsamp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo


def main(nsamp=128):
    np.random.seed(10)                                    # the reference's legacy stream: same seed, same field
    box = CosmoBox(cosmo=default_cosmo, box_scale=(1e3, 1e3, 1e3), nsamp=nsamp, realise_now=False)
    box.realise_density()
    delta_ln = box.lognormal(box.delta_x)
    k, pk, stddev = box.binned_power_spectrum()
    k_ln, pk_ln, _ = box.binned_power_spectrum(delta_x=delta_ln)
    th_k, th_pk = box.theoretical_power_spectrum()
    good = ~np.isnan(pk)
    print("box %d^3: sigma(delta_x) = %.4f, min(log-normal) = %.4f" % (nsamp, np.std(box.delta_x), np.min(delta_ln)))
    for kk, p, e, pl in list(zip(k[good], pk[good], stddev[good], pk_ln[good]))[:8]:
        print("  k = %.4f  P = %10.2f +- %8.2f   theory %10.2f   log-normal %10.2f" % (kk, p, e, np.interp(kk, th_k, th_pk), pl))
    s1, s2 = box.test_parseval()
    return pk


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 128)
