"""CPU: the C ABI library builds, loads, and exports every symbol declared in include/*.h; the
ctypes prototypes cover exactly that list; without a GPU every compute entry fails loudly."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "fastbox_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fb_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from fastbox_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build_library()
    lib = _lib.load()
    names = _declared()
    assert len(names) > 40
    for n in names:
        assert hasattr(lib, n), "libfastbox_hip.so lacks %s" % n
    assert sorted(_lib.SIGNATURES) == names, set(names) ^ set(_lib.SIGNATURES)
    assert lib.fb_version() >= 100


def test_no_cpu_fallback_without_gpu():
    """On a machine without a GPU constructing a box must raise, not compute on the CPU."""
    from fastbox_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from fastbox_amd import CosmoBox, default_cosmo
    with pytest.raises(_lib.FastBoxError):
        CosmoBox(cosmo=default_cosmo, box_scale=1e2, nsamp=16, realise_now=False)
    with pytest.raises(TypeError):                      # argument checks precede the device
        CosmoBox(cosmo=[0.7, 0.3], box_scale=1e2, nsamp=16, realise_now=False)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "fastbox_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
