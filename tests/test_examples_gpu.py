"""The scripts under examples/ run (small grids) and the literal reference idiom gives the device result."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_example_box():
    pk = _load("example_box").main(32)
    assert np.all(np.isfinite(pk[~np.isnan(pk)]))


def test_example_redshift_space_literal_and_device_idioms_agree():
    ex = _load("example_redshift_space")
    a, b = ex.main(32, literal=True), ex.main(32, literal=False)
    assert a.shape == (32, 32, 32) and np.max(np.abs(a - b)) < 1e-3 * np.std(b) + 1e-4


def test_example_endtoend():
    out = _load("example_endtoend").main(32)
    assert out.shape == (32, 32, 32) and np.all(np.isfinite(out))
