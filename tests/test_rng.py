"""CPU: the host model of the device generator (fastbox_amd/rng.py) against the published
Random123 known-answer vectors for Philox4x32-10, the Hermitian planes of the half-spectrum
noise, and basic Box-Muller statistics."""
import numpy as np

from fastbox_amd import rng


def _ph(ctr, key):
    out = rng.philox4x32_10([np.uint64(c) for c in ctr], key)
    return [int(np.asarray(x).ravel()[0]) for x in out]


def test_philox_known_answers():
    # Random123 kat_vectors, "philox4x32 10"
    m = 0xFFFFFFFF
    assert _ph([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _ph([m] * 4, [m] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
               [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_noise_statistics_and_determinism():
    z = rng.half_spectrum_noise(32, seed=7, realisation=3)
    assert z.shape == (32, 32, 17)
    s = np.sqrt(0.5)
    assert abs(z.real.mean()) < 0.02 and abs(z.imag.mean()) < 0.02
    assert abs(z.real.std() - s) < 0.02 and abs(z.imag.std() - s) < 0.02
    assert abs(np.mean(z.real * z.imag)) < 0.02
    assert np.array_equal(z, rng.half_spectrum_noise(32, 7, 3))
    assert not np.array_equal(z, rng.half_spectrum_noise(32, 7, 4))
    z32 = rng.half_spectrum_noise(32, 7, 3, dtype=np.float32)
    assert np.max(np.abs(z32 - z)) < 1e-5
    assert np.array_equal(rng.half_spectrum_noise(32, 7, 3, planes=[0, 5, 16]), z[:, :, [0, 5, 16]])


def test_self_mirrored_planes_are_hermitian():
    """k_z = 0 and N/2: z(-k) = conj z(k), the four self-mirrored modes real with unit variance like every
    other mode -- the half spectrum is the transform of a real field as it stands."""
    N = 32
    z = rng.half_spectrum_noise(N, seed=11, realisation=0)
    m = (-np.arange(N)) % N
    for iz in (0, N // 2):
        P = z[:, :, iz]
        assert np.array_equal(P, np.conj(P[m][:, m]))
        for ix in (0, N // 2):
            for iy in (0, N // 2):
                assert P[ix, iy].imag == 0.0
    full = np.fft.irfftn(z, s=(N, N, N), axes=(0, 1, 2))
    # the same field from the full Hermitian cube: nothing is dropped by irfftn's projection
    cube = np.zeros((N, N, N), dtype=complex)
    cube[:, :, :N // 2 + 1] = z
    cube[:, :, N // 2 + 1:] = np.conj(z[m][:, m][:, :, 1:N // 2][:, :, ::-1])
    assert np.max(np.abs(np.fft.ifftn(cube).imag)) < 1e-15
    assert np.max(np.abs(np.fft.ifftn(cube).real - full)) < 1e-15
    # variance of the planes = variance of the other modes (many realisations)
    v0 = np.mean([np.mean(np.abs(rng.half_spectrum_noise(16, 3, r, planes=[0, 8])) ** 2) for r in range(40)])
    assert abs(v0 - 1.0) < 0.05


def test_single_normal_streams():
    a = rng.stream_normals(1001, 1, seed=5)
    assert a.shape == (1001,) and abs(a.mean()) < 0.15 and abs(a.std() - 1) < 0.1
    assert np.array_equal(a[:8], rng.stream_normals(8, 1, seed=5))
    assert not np.array_equal(a[:8], rng.stream_normals(8, 3, seed=5)[:8])
    assert np.array_equal(rng.los_noise(8, 5).ravel(), rng.stream_normals(512, 1, 5))
