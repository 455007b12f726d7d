"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Loads the *reference's own* ``fastbox/box.py`` by file path (read-only, from
/root/reference, which exists only in the build container) so that the numpy
restatement in ``oracle/box_oracle.py`` can be pinned against the reference
itself and so that ``oracle/make_golden.py`` can emit golden vectors.

``import fastbox`` fails here (pyccl / skimage / IPython are absent), but
``box.py`` alone needs only numpy, scipy, pylab and ``pyccl``; the latter is
replaced by a module object that forwards to ``fastbox_amd.cosmology`` (the
same closed-form provider the product uses when pyccl is missing), so that
reference and product see identical P(k), E(a), f(a), D(a), chi(a).
"""
import importlib.util
import os
import sys
import types

REFERENCE_BOX = "/root/reference/fastbox/box.py"


def reference_available():
    return os.path.exists(REFERENCE_BOX)


def load_reference_box():
    """Return the reference ``box`` module (CosmoBox, default_cosmo)."""
    if not reference_available():
        raise RuntimeError("reference sources are not present on this machine")
    sys.dont_write_bytecode = True
    import matplotlib
    matplotlib.use("Agg")
    import scipy.integrate
    if not hasattr(scipy.integrate, "simps"):       # box.py:680 predates scipy 1.14
        scipy.integrate.simps = scipy.integrate.simpson

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from fastbox_amd import cosmology as cosmo_mod

    shim = types.ModuleType("pyccl")
    for name in ("Cosmology", "nonlin_matter_power", "linear_matter_power",
                 "h_over_h0", "growth_rate", "growth_factor",
                 "comoving_angular_distance"):
        setattr(shim, name, getattr(cosmo_mod, name))
    had = sys.modules.get("pyccl")
    sys.modules["pyccl"] = shim
    try:
        spec = importlib.util.spec_from_file_location("_fastbox_reference_box",
                                                      REFERENCE_BOX)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if had is None:
            sys.modules.pop("pyccl", None)
        else:
            sys.modules["pyccl"] = had
    return mod


def load_reference_module(relpath, name):
    """Load another single-file module of the reference package (e.g. 'fastbox/noise.py') with the same
    stand-in pyccl.  Only modules whose imports are available here work (noise.py, foregrounds.py: numpy,
    scipy.ndimage, pylab; healpy is optional there)."""
    path = os.path.join("/root/reference", relpath)
    if not os.path.exists(path):
        raise RuntimeError("reference sources are not present on this machine")
    sys.dont_write_bytecode = True
    import warnings
    import matplotlib
    matplotlib.use("Agg")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from fastbox_amd import cosmology as cosmo_mod
    shim = types.ModuleType("pyccl")
    for nm in ("Cosmology", "nonlin_matter_power", "linear_matter_power", "h_over_h0", "growth_rate",
               "growth_factor", "comoving_angular_distance"):
        setattr(shim, nm, getattr(cosmo_mod, nm))
    had = sys.modules.get("pyccl")
    sys.modules["pyccl"] = shim
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            spec = importlib.util.spec_from_file_location(name, path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
    finally:
        if had is None:
            sys.modules.pop("pyccl", None)
        else:
            sys.modules["pyccl"] = had
    return mod


def load_reference_filters():
    """fastbox/filters.py does `from .foregrounds import ...`: give it a stand-in parent package whose __path__ is
    the reference directory (fastbox/__init__.py itself is NOT executed: it imports modules that are absent here)."""
    if not os.path.exists("/root/reference/fastbox/filters.py"):
        raise RuntimeError("reference sources are not present on this machine")
    sys.dont_write_bytecode = True
    import importlib
    import warnings
    import matplotlib
    matplotlib.use("Agg")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from fastbox_amd import cosmology as cosmo_mod
    shim = types.ModuleType("pyccl")
    for nm in ("Cosmology", "nonlin_matter_power", "linear_matter_power", "h_over_h0", "growth_rate",
               "growth_factor", "comoving_angular_distance"):
        setattr(shim, nm, getattr(cosmo_mod, nm))
    had = sys.modules.get("pyccl")
    sys.modules["pyccl"] = shim
    pkg = types.ModuleType("_fastbox_reference_pkg")
    pkg.__path__ = ["/root/reference/fastbox"]
    sys.modules["_fastbox_reference_pkg"] = pkg
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mod = importlib.import_module("_fastbox_reference_pkg.filters")
    finally:
        if had is None:
            sys.modules.pop("pyccl", None)
        else:
            sys.modules["pyccl"] = had
    return mod
