"""
Self-contained background/P(k) provider with the slice of the ``pyccl`` call
surface that the density-field hot path touches.

The reference obtains P(k), H(a)/H0, the growth rate/factor and the comoving
distance from ``pyccl`` (reference call sites: fastbox/box.py:62, 163, 165,
280-281, 345, 406, 781, 820, 851, 889, 896).  ``pyccl`` is a third-party
dependency that is not vendored in the reference and is not installed in this
image, so the values it would return are *parity unpinned* (SURVEY.md par. 8c).
This module supplies the same function names with a closed-form flat-LCDM
model so that ``CosmoBox(default_cosmo)`` works offline:

  * linear P(k): Eisenstein & Hu (1998) zero-baryon ("no-wiggle") transfer
    function, primordial tilt ``n_s``, normalised to ``sigma8`` with an
    8 Mpc/h top-hat;
  * ``nonlin_matter_power`` == linear (no halofit correction yet);
  * E(a), D(a), f(a), chi(a) for flat LCDM without radiation.

If the real ``pyccl`` is importable, ``fastbox_amd.box`` uses it instead.
Everything downstream of these scalars/tables is what the parity tests pin.
"""
import numpy as np

C_KMS = 299792.458  # speed of light, km/s

_TRAPZ = getattr(np, "trapezoid", None) or np.trapz


class Cosmology(object):
    """Parameter holder with ``cosmo['h']``-style access (pyccl idiom)."""

    _defaults = dict(Omega_c=0.25, Omega_b=0.05, h=0.7, n_s=0.95, sigma8=0.8,
                     transfer_function='eisenstein_hu', T_CMB=2.7255)

    def __init__(self, **params):
        unknown = [k for k in params if k not in self._defaults
                   and k not in ('matter_power_spectrum', 'Omega_k', 'A_s',
                                 'Neff', 'm_nu', 'w0', 'wa')]
        if unknown:
            raise TypeError("unknown cosmological parameter(s): %s" % unknown)
        self._p = dict(self._defaults)
        self._p.update(params)
        self._p['Omega_m'] = self._p['Omega_c'] + self._p['Omega_b']
        self._p['Omega_l'] = 1.0 - self._p['Omega_m']
        self._norm = None

    def __getitem__(self, key):
        return self._p[key]

    # -- linear power at a=1, un-normalised ---------------------------------
    def _transfer_nowiggle(self, k):
        """EH98 eqs. 26-31; ``k`` in 1/Mpc."""
        p = self._p
        h = p['h']
        om, ob = p['Omega_m'], p['Omega_b']
        omh2, obh2, fb = om * h * h, ob * h * h, ob / om
        theta = p['T_CMB'] / 2.7
        s = 44.5 * np.log(9.83 / omh2) / np.sqrt(1.0 + 10.0 * obh2 ** 0.75)
        ag = 1.0 - 0.328 * np.log(431.0 * omh2) * fb \
            + 0.38 * np.log(22.3 * omh2) * fb * fb
        gamma_eff = om * h * (ag + (1.0 - ag) / (1.0 + (0.43 * k * s) ** 4))
        q = (k / h) * theta * theta / gamma_eff
        L0 = np.log(2.0 * np.e + 1.8 * q)
        C0 = 14.2 + 731.0 / (1.0 + 62.5 * q)
        return L0 / (L0 + C0 * q * q)

    def _pk_shape(self, k):
        return k ** self._p['n_s'] * self._transfer_nowiggle(k) ** 2

    def _amplitude(self):
        if self._norm is None:
            R = 8.0 / self._p['h']
            lk = np.linspace(np.log(1e-5), np.log(1e2), 20001)
            k = np.exp(lk)
            x = k * R
            w = 3.0 * (np.sin(x) - x * np.cos(x)) / x ** 3
            integrand = k ** 3 * self._pk_shape(k) * w * w / (2.0 * np.pi ** 2)
            self._norm = self._p['sigma8'] ** 2 / _TRAPZ(integrand, lk)
        return self._norm


def _E(cosmo, a):
    a = np.asarray(a, dtype=np.float64)
    return np.sqrt(cosmo['Omega_m'] * a ** -3 + cosmo['Omega_l'])


def h_over_h0(cosmo, a):
    """E(a) = H(a)/H0."""
    return _E(cosmo, a)


def _growth_integral(cosmo, a):
    # I(a) = int_0^a da' / (a' E(a'))^3, substitution a' = a u^(2/5) tames the
    # a'^(3/2) behaviour of the integrand at the origin.
    u = np.linspace(0.0, 1.0, 4001)[1:]
    ap = a * u ** 0.4
    jac = 0.4 * a * u ** -0.6
    y = jac / (ap * _E(cosmo, ap)) ** 3
    return _TRAPZ(np.concatenate(([y[0]], y)), np.concatenate(([0.0], u)))


def _growth_unnorm(cosmo, a):
    return 2.5 * cosmo['Omega_m'] * _E(cosmo, a) * _growth_integral(cosmo, a)


def growth_factor(cosmo, a):
    """Linear growth D(a), normalised to D(1) = 1."""
    return float(_growth_unnorm(cosmo, float(a)) / _growth_unnorm(cosmo, 1.0))


def growth_rate(cosmo, a):
    """f = dlnD/dlna."""
    a = float(a)
    E = float(_E(cosmo, a))
    dlnE = -1.5 * cosmo['Omega_m'] * a ** -3 / (E * E)
    return float(dlnE + 1.0 / (a * a * E ** 3 * _growth_integral(cosmo, a)))


def comoving_angular_distance(cosmo, a):
    """Flat-space comoving distance to scale factor ``a``, in Mpc."""
    a = float(a)
    if a >= 1.0:
        return 0.0
    aa = np.linspace(a, 1.0, 2049)
    return float((C_KMS / (100.0 * cosmo['h']))
                 * _TRAPZ(1.0 / (aa * aa * _E(cosmo, aa)), aa))


def linear_matter_power(cosmo, k, a):
    """P_lin(k, a) in Mpc^3 for ``k`` in 1/Mpc; NaN at k = 0 (the reference
    relies on ``nan_to_num`` to zero the DC mode, fastbox/box.py:167)."""
    k = np.asarray(k, dtype=np.float64)
    D = growth_factor(cosmo, a)
    out = np.full(k.shape, np.nan)
    good = k > 0.0
    out[good] = cosmo._amplitude() * cosmo._pk_shape(k[good]) * D * D
    return out


def nonlin_matter_power(cosmo, k, a):
    """Stand-in: identical to the linear spectrum (no halofit)."""
    return linear_matter_power(cosmo, k, a)
