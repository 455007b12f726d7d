"""Beam convolution on the device (fastbox_amd/beams.py, fb_beam_convolve) against vectors captured from the
reference's fastbox/beams.py (tests/golden/beam_*.npz, oracle/make_golden_beams.py) and against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import beam_oracle as bo       # noqa: E402
from oracle import standin                  # noqa: E402


def g_beam_of(model):
    return model.beam_cube()


def _case(golden_dir, name, precision):
    from fastbox_amd import BeamModel, CosmoBox
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    box = CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=tuple(g["box_scale"]), nsamp=int(g["N"]),
                   redshift=float(g["redshift"]), realise_now=False, precision=precision)
    cube = g["beam"]

    class FixtureBeam(BeamModel):
        def beam_cube(self, pol=None):
            return cube

    return g, box, FixtureBeam(box)


@pytest.mark.parametrize("name", ["beam_n16", "beam_n32"])
@pytest.mark.parametrize("precision,tol", [("f64", 1e-12), ("f32", 3e-6)])
def test_beam_convolution_matches_reference_vectors(golden_dir, name, precision, tol):
    from fastbox_amd import BeamModel
    g, box, beam = _case(golden_dir, name, precision)
    field = g["field"]
    scale = np.max(np.abs(g["conv_fft"]))
    out = beam.convolve_fft(field)
    assert out.shape == field.shape and out.dtype == np.float64
    assert np.max(np.abs(np.asarray(out) - g["conv_fft"])) < tol * scale
    again = beam.convolve_fft(2. * field)              # same beam object: its transform is reused
    assert beam._beam_k[0] is g_beam_of(beam) and np.max(np.abs(np.asarray(again) - 2. * g["conv_fft"])) < 2 * tol * scale
    # a device cube in, the base class's uniform beam (beams.py:26-38)
    dev_field = box.engine.upload(field, "real")
    assert np.max(np.abs(np.asarray(BeamModel(box).convolve_fft(dev_field)) - g["conv_fft_uniform"])) < tol * scale
    if "conv_real" in g.files:
        assert np.max(np.abs(np.asarray(beam.convolve_real(field)) - g["conv_real"])) < tol * scale
    assert np.array_equal(BeamModel(box).beam_value(np.zeros(3), np.ones(3), np.ones(3)), np.ones(3))
    with pytest.raises(AssertionError):
        BeamModel(box).beam_value(np.zeros(3), np.ones(2), np.ones(3))


@pytest.mark.parametrize("precision,tol", [("f64", 1e-12), ("f32", 3e-6)])
def test_beam_convolution_of_a_realised_box_against_the_oracle(precision, tol):
    """128^3: the lazy density field goes straight in; both convolutions against oracle/beam_oracle.py."""
    from fastbox_amd import BeamModel, CosmoBox, default_cosmo
    np.random.seed(8)
    box = CosmoBox(cosmo=default_cosmo, box_scale=(2e3, 2e3, 1e3), nsamp=128, redshift=0.6, realise_now=False,
                   precision=precision)
    dx = box.realise_density()
    ang_x, ang_y = box.pixel_array()
    cube = bo.test_beam_cube(ang_x, ang_y, box.freq_array(), fwhm_deg=0.2 * (ang_x[-1] - ang_x[0]))

    class FixtureBeam(BeamModel):
        def beam_cube(self, pol=None):
            return cube

    beam = FixtureBeam(box)
    host = np.asarray(dx)
    want = bo.convolve_fft(cube, host)
    scale = np.max(np.abs(want))
    assert np.max(np.abs(np.asarray(beam.convolve_fft(dx)) - want)) < tol * scale
    want = bo.convolve_real(cube, host)
    assert np.max(np.abs(np.asarray(beam.convolve_real(dx)) - want)) < tol * np.max(np.abs(want))
    # a smooth beam lowers the variance and leaves the mean (the normalisation) alone, away from the zero-padded edge
    sm = np.asarray(beam.convolve_real(dx))
    assert np.std(sm) < np.std(host) and abs(np.mean(sm) - np.mean(host)) < 1e-4 * np.std(host) + 1e-6
