"""CPU: the host model of the device generator (fastbox_amd/rng.py) against the published
Random123 known-answer vectors for Threefry4x32-20, and basic Box-Muller statistics."""
import numpy as np

from fastbox_amd import rng


def _tf(ctr, key):
    out = rng.threefry4x32_20([np.uint64(c) for c in ctr], key)
    return [int(np.asarray(x).ravel()[0]) for x in out]


def test_threefry_known_answers():
    m = 0xFFFFFFFF
    assert _tf([0] * 4, [0] * 4) == [0x9c6ca96a, 0xe17eae66, 0xfc10ecd4, 0x5256a7d8]
    assert _tf([m] * 4, [m] * 4) == [0x2a881696, 0x57012287, 0xf6c7446e, 0xa16a6732]
    assert _tf([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
               [0xa4093822, 0x299f31d0, 0x082efa98, 0xec4e6c89]) == [0x59cd1dbb, 0xb8879579, 0x86b5d00c, 0xac8b6d84]


def test_noise_statistics_and_determinism():
    z = rng.half_spectrum_noise(32, seed=7, realisation=3)
    assert z.shape == (32, 32, 17)
    assert abs(z.real.mean()) < 0.02 and abs(z.imag.mean()) < 0.02
    assert abs(z.real.std() - 1) < 0.02 and abs(z.imag.std() - 1) < 0.02
    assert abs(np.mean(z.real * z.imag)) < 0.02
    assert np.array_equal(z, rng.half_spectrum_noise(32, 7, 3))
    assert not np.array_equal(z, rng.half_spectrum_noise(32, 7, 4))
    z32 = rng.half_spectrum_noise(32, 7, 3, dtype=np.float32)
    assert np.max(np.abs(z32 - z)) < 1e-5
