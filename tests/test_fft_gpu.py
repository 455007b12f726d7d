"""GPU: the HIP 3-D FFT passes (through the C ABI) against numpy.fft on the same inputs.
Tolerances: relative L2/Linf error <= 2e-6 for fp32 storage, <= 1e-13 for fp64."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = {"f32": 3e-6, "f64": 2e-13}


def _engine(N, precision, L=(1e3, 1e3, 1e3)):
    from fastbox_amd.device import Engine
    m = (N * np.fft.fftfreq(N, 1.)).astype("i").astype(np.float64)
    axis2 = np.concatenate([(m / l) ** 2. for l in L])
    ksc = np.concatenate([m * (2. * np.pi / l) for l in L])
    kpar = 2. * np.pi * m / L[2]
    z = np.linspace(-0.5 * L[2], 0.5 * L[2], N)
    return Engine(N, L, axis2, ksc, kpar, z, precision=precision)


def _err(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("N", [16, 32, 64, 128, 256])
def test_c2c_roundtrip_and_values(N, precision):
    from fastbox_amd.device import FULL
    eng = _engine(N, precision)
    rng = np.random.RandomState(N)
    x = rng.normal(size=(N, N, N)) + 1j * rng.normal(size=(N, N, N))
    d = eng.upload(x, FULL)
    xin = d.host().copy()                 # what the device actually holds (fp32-rounded for f32)
    fwd = eng.fft_c2c(d, -1)
    ref = np.fft.fftn(xin)
    assert _err(fwd.host(), ref) < TOL[precision]
    back = eng.fft_c2c(fwd, +1, 1.0 / N ** 3)
    assert _err(back.host(), xin) < TOL[precision]
    inv = eng.fft_c2c(d, +1, 1.0 / N ** 3)
    assert _err(inv.host(), np.fft.ifftn(xin)) < TOL[precision]


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("N", [16, 32, 64, 128, 256])
def test_r2c_c2r(N, precision):
    from fastbox_amd.device import REAL
    eng = _engine(N, precision)
    rng = np.random.RandomState(100 + N)
    x = rng.normal(size=(N, N, N))
    d = eng.upload(x, REAL)
    xin = d.host().copy()
    half = eng.fft_r2c(d)
    ref = np.fft.fftn(xin)
    raw = eng.download_half_raw(half)[:, :, :N // 2 + 1]
    assert _err(raw, ref[:, :, :N // 2 + 1]) < TOL[precision]
    assert _err(half.host(), ref) < TOL[precision]           # Hermitian extension
    back = eng.fft_c2r(half)
    assert _err(back.host(), xin) < TOL[precision]
    # the half spectrum survives a non-destroying c2r
    assert _err(eng.download_half_raw(half)[:, :, :N // 2 + 1], ref[:, :, :N // 2 + 1]) < TOL[precision]


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_c2r_projects_non_hermitian_planes(precision):
    """Re ifftn(X) for a half spectrum whose k_z = 0, N/2 planes are not Hermitian."""
    from fastbox_amd.device import FULL
    N = 32
    eng = _engine(N, precision)
    rng = np.random.RandomState(7)
    X = rng.normal(size=(N, N, N)) + 1j * rng.normal(size=(N, N, N))
    herm = 0.5 * (X + np.conj(np.roll(X[::-1, ::-1, ::-1], 1, axis=(0, 1, 2))))
    mixed = herm.copy()
    mixed[:, :, 0] = X[:, :, 0]
    mixed[:, :, N // 2] = X[:, :, N // 2]
    half = eng.crop_full(eng.upload(mixed, FULL))
    out = eng.fft_c2r(half).host()
    ref = np.fft.ifftn(X).real
    assert _err(out, ref) < 10 * TOL[precision]


def test_unsupported_size_fails_loudly():
    from fastbox_amd._lib import FastBoxError
    for N in (22, 1026, 25):           # a prime factor 11, beyond the plain passes' range, odd (24, 48, ...: tests/test_generic_grid_gpu.py)
        with pytest.raises(FastBoxError):
            _engine(N, "f32")


@pytest.mark.parametrize("precision,tol", [("f32", 2e-6), ("f64", 1e-12)])
def test_fused_exponential_of_the_forward_transform(precision, tol):
    """r2c with pre_exp (the log-normal fusion): the transform of exp(x) against numpy on the same stored x, and a
    column whose every line is constant, where the k_z = 0 mode of a line is exactly N exp(x): the device exponential
    (fb_exp: 2^(x log2 e) with a first-order correction of the product's rounding) over four decades of magnitude."""
    from fastbox_amd.device import REAL
    N = 32
    eng = _engine(N, precision)
    rng = np.random.RandomState(3)
    x = 1.5 * rng.normal(size=(N, N, N))
    x[:, 0, :] = np.linspace(-9., 9., N)[:, None]                  # constant along z: e^-9 ... e^9
    d = eng.upload(x, REAL)
    xin = d.host().copy()
    got = eng.fft_r2c(d, pre_exp=True).host()
    want = np.fft.fftn(np.exp(xin))
    assert _err(got, want) < tol
    # the constant lines: sum over y of the first transform's k_z = 0 mode is the only place they enter with weight
    # N exp(x); compare the (k_x, k_y = all, k_z = 0) plane, which is linear in exp(x[:, 0, 0]) up to the other rows
    assert np.max(np.abs(got[:, :, 0] - want[:, :, 0])) < tol * np.max(np.abs(want[:, :, 0]))
