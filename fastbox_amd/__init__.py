"""
fastbox_amd -- MI355X-native density-field hot path of philbull/FastBox.

    from fastbox_amd import CosmoBox, default_cosmo      # as fastbox.box

Importing the package does not need a GPU; constructing a ``CosmoBox`` does
(the HIP library is loaded and a plan is created on the device; there is no
CPU fallback).
"""
from .box import CosmoBox, default_cosmo          # noqa: F401
from .transfer import BeamHighpass, Wedge         # noqa: F401
from .device import DeviceArray                   # noqa: F401
from .sky import ForegroundModel, NoiseModel      # noqa: F401
from . import filters                             # noqa: F401
from . import montecarlo                          # noqa: F401
from .beams import BeamModel                      # noqa: F401

__version__ = "0.1.0"
