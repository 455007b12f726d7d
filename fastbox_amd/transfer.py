"""
Transfer functions T(k_perp, k_par) for ``CosmoBox.apply_transfer_fn``.

The reference takes an arbitrary Python callable (fastbox/box.py:356-381).  A
callable cannot run on the GPU, so the common shapes are provided as objects
that are *both* ordinary numpy callables (the reference itself accepts them,
which is how the parity tests compare the two) *and* carry the parameters of
the on-device evaluator (FB_FILT_* in include/fastbox_hip.h).  Any other
callable is evaluated once on the host grid and uploaded as a multiplier
table.
"""
import numpy as np

from ._lib import FB_FILT_BEAM_HIGHPASS, FB_FILT_WEDGE


class DeviceFilter(object):
    """Base: subclasses define kind, params and __call__(k_perp, k_par).
    `even_in_kpar`: T(k_perp, -k_par) == T(k_perp, k_par), so a Hermitian field
    stays Hermitian and the real-to-complex fast path applies."""
    kind = None
    params = (0.0, 0.0, 0.0, 0.0)
    even_in_kpar = True


class BeamHighpass(DeviceFilter):
    """(1 - exp(-0.5 (|k_par|/kpar0)^power)) * exp(-0.5 (k_perp/kperp0)^2).

    ``kpar0=None`` drops the high-pass factor, ``kperp0=None`` the beam factor.
    power=2 with both scales is the filter of fastbox/tests/test_box.py:88-90;
    kperp0=None, power=3 is examples/example_endtoend.py:133.
    """
    kind = FB_FILT_BEAM_HIGHPASS

    def __init__(self, kpar0=None, kperp0=None, power=2.0):
        self.kpar0, self.kperp0, self.power = kpar0, kperp0, float(power)
        self.params = (float(kpar0 or 0.0), float(kperp0 or 0.0), self.power, 0.0)

    def __call__(self, k_perp, k_par):
        out = 1.0
        if self.kpar0:
            out = out * (1. - np.exp(-0.5 * (np.abs(k_par) / self.kpar0) ** self.power))
        if self.kperp0:
            out = out * np.exp(-0.5 * (k_perp / self.kperp0) ** 2.)
        return out


class Wedge(DeviceFilter):
    """Foreground-wedge mask: 0 where |k_par| < slope * k_perp + kpar_min, else 1."""
    kind = FB_FILT_WEDGE

    def __init__(self, slope, kpar_min=0.0):
        self.slope, self.kpar_min = float(slope), float(kpar_min)
        self.params = (self.slope, self.kpar_min, 0.0, 0.0)

    def __call__(self, k_perp, k_par):
        return np.where(np.abs(k_par) < self.slope * k_perp + self.kpar_min, 0.0, 1.0)
