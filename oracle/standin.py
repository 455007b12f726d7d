"""TEST INFRASTRUCTURE ONLY.  Inputs shared by the oracle, the golden-vector
generator and the GPU parity tests: the stand-in cosmology scalars/P(k) (the same
closed-form provider the product uses when pyccl is absent) and the transfer
functions the reference's own tests and examples use."""
import numpy as np

from fastbox_amd import cosmology as cosmo_mod

DEFAULT_COSMO = dict(Omega_c=0.25, Omega_b=0.05, h=0.7, n_s=0.95, sigma8=0.8,
                     transfer_function='eisenstein_hu')


def cosmology():
    return cosmo_mod.Cosmology(**DEFAULT_COSMO)


def pk_fn(cosmo, a=1.0):
    return lambda k: cosmo_mod.nonlin_matter_power(cosmo, k=k, a=a)


def velocity_fac(cosmo, a):
    """box.py:280-281."""
    return 100. * cosmo['h'] * cosmo_mod.h_over_h0(cosmo, a=a) * cosmo_mod.growth_rate(cosmo, a=a) * a


def hubble(cosmo, a):
    """box.py:406."""
    return 100. * cosmo['h'] * cosmo_mod.h_over_h0(cosmo, a)


def beam_highpass(k_perp, k_par):
    """fastbox/tests/test_box.py:88-90."""
    return (1. - np.exp(-0.5 * (k_par / 0.001) ** 2.)) * np.exp(-0.5 * (k_perp / 0.1) ** 2.)


def highpass3(kperp, kpar):
    """examples/example_endtoend.py:133."""
    return 1. - np.exp(-0.5 * (np.abs(kpar) / 0.009) ** 3.)


def wedge03(k_perp, k_par):
    """Foreground-wedge cut of BASELINE.json configs[2] (ours; the reference has no wedge filter, only user callables):
    0 where |k_par| < 0.3 k_perp, else 1 -- fastbox_amd.Wedge(slope=0.3) as a reference-style callable."""
    return np.where(np.abs(k_par) < 0.3 * k_perp, 0., 1.)
