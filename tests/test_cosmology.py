"""CPU: known answers for the built-in P(k) / background provider (fastbox_amd/cosmology.py), which stands in for
``pyccl`` at the reference's call sites (fastbox/box.py:62, 163-165, 280-281, 345, 406, 781).

PARITY UNPINNED: pyccl is not installed here, so none of this is a comparison with the reference's provider.  What is
asserted are the published properties of the formulae the module says it implements -- Eisenstein & Hu 1998 (EH98),
Smith et al. 2003 / Takahashi et al. 2012 (halofit), flat-LCDM background -- each recomputed by an independent route
(other quadrature, finite differences, closed forms of limiting cases)."""
import numpy as np
from scipy import integrate

from fastbox_amd import cosmology as cm

DEFAULT = dict(Omega_c=0.25, Omega_b=0.05, h=0.7, n_s=0.95, sigma8=0.8, transfer_function='eisenstein_hu')


def _cosmo(**kw):
    p = dict(DEFAULT)
    p.update(kw)
    return cm.Cosmology(**p)


def test_eh98_transfer_function_limits():
    for tf in ("eisenstein_hu", "eisenstein_hu_nowiggles"):
        c = _cosmo(transfer_function=tf)
        T = c._transfer_wiggles if tf == "eisenstein_hu" else c._transfer_nowiggle
        # T -> 1 on scales far beyond the horizon at equality (EH98 eq. 16 normalisation)
        assert abs(T(np.array([1e-6]))[0] - 1.0) < 1e-6 and abs(T(np.array([1e-4]))[0] - 1.0) < 2e-3
        # monotone envelope, positive, and the small-scale law T ~ ln(k) / k^2 (eqs. 19, 29)
        k = np.logspace(-4, 2, 400)
        assert np.all(T(k) > 0) and T(k)[-1] < 1e-5
        hi = T(np.array([50.0, 100.0]))
        slope = np.log(hi[1] / hi[0]) / np.log(2.0)
        assert -2.0 < slope < -1.7                       # k^-2 times a slowly growing logarithm
    # primordial tilt: P ~ k^n_s as k -> 0
    c = _cosmo()
    k = np.array([1e-5, 2e-5])
    p = cm.linear_matter_power(c, k, 1.0)
    assert abs(np.log(p[1] / p[0]) / np.log(2.0) - 0.95) < 1e-3
    assert np.isnan(cm.linear_matter_power(c, np.array([0.0]), 1.0))[0]     # k = 0 -> NaN, as pyccl (box.py:167 relies on it)


def test_eh98_scales_against_the_papers_fits():
    c = _cosmo()
    zeq, keq, zd, s, ksilk = c._eh98_scales()
    omh2, obh2, th = 0.30 * 0.49, 0.05 * 0.49, 2.7255 / 2.7
    assert abs(zeq - 2.5e4 * omh2 / th ** 4) < 1e-9 and 3400 < zeq < 3700          # eq. 2
    assert abs(keq - 0.0746 * omh2 / th ** 2) < 1e-12                                # eq. 3, 1/Mpc
    assert 1000 < zd < 1080                                                         # drag epoch, eq. 4
    # eq. 26: s ~ 44.5 ln(9.83 / Om h^2) / sqrt(1 + 10 (Ob h^2)^(3/4)) Mpc, "accurate to ~2 %" for Ob h^2 >~ 0.0125
    s_fit = 44.5 * np.log(9.83 / omh2) / np.sqrt(1.0 + 10.0 * obh2 ** 0.75)
    assert abs(s / s_fit - 1.0) < 0.02 and 140 < s < 155
    assert 0.08 < ksilk < 0.2                                                       # Silk scale, eq. 7, 1/Mpc
    # acoustic oscillations: the ratio to the zero-baryon form wiggles with period ~ 2 pi / s in k
    k = np.linspace(0.03, 0.3, 4000)
    r = (c._transfer_wiggles(k) / c._transfer_nowiggle(k)) ** 2
    r = r - np.convolve(r, np.ones(801) / 801, mode="same")
    inner = slice(500, 3500)
    up = np.nonzero((r[inner][1:-1] > r[inner][:-2]) & (r[inner][1:-1] > r[inner][2:]))[0]
    period = np.mean(np.diff(k[inner][1:-1][up]))
    assert abs(period / (2 * np.pi / s) - 1.0) < 0.1
    assert 0.02 < np.max(np.abs(r[inner])) < 0.15                                    # a few per cent of power


def test_sigma8_normalisation_by_independent_quadrature():
    for tf, s8 in (("eisenstein_hu", 0.8), ("eisenstein_hu_nowiggles", 0.65)):
        c = _cosmo(transfer_function=tf, sigma8=s8)
        R = 8.0 / 0.7

        def f(lnk):
            k = np.exp(lnk)
            x = k * R
            w = 3.0 * (np.sin(x) - x * np.cos(x)) / x ** 3
            return k ** 3 * cm.linear_matter_power(c, np.array([k]), 1.0)[0] * w * w / (2 * np.pi ** 2)
        val, err = integrate.quad(f, np.log(1e-6), np.log(50.0), limit=400, epsrel=1e-8)
        assert abs(np.sqrt(val) - s8) < 1e-4 * s8


def test_halofit_scale_slope_and_curvature_are_what_their_definitions_say():
    """Smith et al. 2003 eqs. C5-C8: sigma^2(R) with a Gaussian filter, sigma(1/k_sigma) = 1,
    n_eff = -3 - dln sigma^2/dln R, C = -d^2 ln sigma^2 / dln R^2, all at R = 1/k_sigma."""
    c = _cosmo()
    ksig, neff, C = cm._halofit_params(c, 1.0)

    def lns2(lnR):
        R = np.exp(lnR)
        f = lambda lnk: np.exp(3 * lnk) * cm.linear_matter_power(c, np.array([np.exp(lnk)]), 1.0)[0] / (2 * np.pi ** 2) \
            * np.exp(-(np.exp(lnk) * R) ** 2)
        return np.log(integrate.quad(f, np.log(1e-6), np.log(60.0 / R), limit=400, epsrel=1e-10)[0])
    l0, h = -np.log(ksig), 0.02
    assert abs(lns2(l0)) < 2e-4                                        # sigma(1/k_sigma) = 1
    d1 = (lns2(l0 + h) - lns2(l0 - h)) / (2 * h)
    d2 = (lns2(l0 + h) - 2 * lns2(l0) + lns2(l0 - h)) / h ** 2
    assert abs((-3.0 - d1) - neff) < 2e-3 and abs(-d2 - C) < 5e-3
    assert 0.2 < ksig < 0.6 and -2.2 < neff < -1.4 and 0.2 < C < 0.5    # LCDM, sigma8 = 0.8, z = 0 (Takahashi+12 fig. 1 range)
    # earlier times: non-linearity sets in at smaller scales
    assert cm._halofit_params(c, 0.5)[0] > 2 * ksig


def test_halofit_limits():
    c = _cosmo()
    k = np.array([1e-3, 1e-2, 0.1, 1.0, 10.0])
    lin, nl = cm.linear_matter_power(c, k, 1.0), cm.nonlin_matter_power(c, k, 1.0)
    assert abs(nl[0] / lin[0] - 1) < 1e-3 and abs(nl[1] / lin[1] - 1) < 1e-2        # linear on large scales
    assert nl[3] / lin[3] > 5 and nl[4] / lin[4] > 20                                # strongly boosted past k_sigma
    kf = np.logspace(-3, 1, 200)
    rf = cm.nonlin_matter_power(c, kf, 1.0) / cm.linear_matter_power(c, kf, 1.0)
    assert rf.min() > 0.97 and np.all(np.diff(rf[kf > 0.15]) > 0)       # a ~1 % quasi-linear dip, then a growing boost
    d2 = k ** 3 * nl / (2 * np.pi ** 2)
    assert 10 < d2[3] < 40                                                            # Delta^2(k = 1/Mpc) ~ 20 at z = 0
    # no non-linear scale (sigma8 tiny): halofit returns the linear spectrum
    weak = _cosmo(sigma8=1e-3)
    assert np.array_equal(cm.nonlin_matter_power(weak, k, 1.0), cm.linear_matter_power(weak, k, 1.0))
    assert np.isnan(cm.nonlin_matter_power(c, np.array([0.0]), 1.0))[0]


def test_background_closed_forms():
    # Einstein-de Sitter: E = a^-3/2, D = a, f = 1, chi = (2 c / H0)(1 - sqrt a)
    eds = _cosmo(Omega_c=0.95, Omega_b=0.05)
    for a in (0.1, 0.5, 1.0):
        assert abs(cm.h_over_h0(eds, a) - a ** -1.5) < 1e-12
        assert abs(cm.growth_factor(eds, a) - a) < 2e-6
        assert abs(cm.growth_rate(eds, a) - 1.0) < 2e-6
    assert abs(cm.comoving_angular_distance(eds, 0.25) - 2 * cm.C_KMS / 70.0 * (1 - 0.5)) < 0.05
    # LCDM: normalisations, the matter-era limit D ~ a, and f ~ Omega_m(a)^0.55 (Linder 2005) to a per cent
    c = _cosmo()
    assert cm.h_over_h0(c, 1.0) == 1.0 and abs(cm.growth_factor(c, 1.0) - 1.0) < 1e-12
    assert abs(cm.growth_factor(c, 0.02) / 0.02 / (cm.growth_factor(c, 0.01) / 0.01) - 1) < 1e-4
    for a in (0.3, 0.6, 1.0):
        om_a = 0.3 * a ** -3 / cm.h_over_h0(c, a) ** 2
        assert abs(cm.growth_rate(c, a) / om_a ** 0.55 - 1) < 0.01
        h = 1e-4                                       # f = dlnD/dlna by finite differences of growth_factor
        fd = (np.log(cm.growth_factor(c, a * (1 + h))) - np.log(cm.growth_factor(c, a * (1 - h)))) / (2 * h)
        assert abs(fd - cm.growth_rate(c, a)) < 1e-5
    chi = integrate.quad(lambda z: cm.C_KMS / (70.0 * np.sqrt(0.3 * (1 + z) ** 3 + 0.7)), 0.0, 1.0)[0]
    assert abs(cm.comoving_angular_distance(c, 0.5) / chi - 1) < 1e-6
    assert cm.comoving_angular_distance(c, 1.0) == 0.0


def test_pyccl_call_surface_the_reference_uses():
    """box.py touches: Cosmology(**dict), cosmo['h'], isinstance(cosmo, ccl.Cosmology), and the six functions."""
    c = cm.Cosmology(**DEFAULT)
    assert isinstance(c, cm.Cosmology) and c['h'] == 0.7 and c['sigma8'] == 0.8 and c['Omega_c'] == 0.25
    for name in ("nonlin_matter_power", "linear_matter_power", "h_over_h0", "growth_rate", "growth_factor",
                 "comoving_angular_distance"):
        assert callable(getattr(cm, name))
    try:
        cm.Cosmology(Omega_x=1.0)
    except TypeError:
        pass
    else:
        raise AssertionError("unknown parameters must be refused")
