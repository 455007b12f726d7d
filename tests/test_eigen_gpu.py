"""GPU: fb_leading_eigenvectors (cyclic two-sided Jacobi in fp64 on the device; the eigen-decomposition step of
fastbox/filters.py:161-169) against numpy.linalg.eigh on the same matrix, through the C ABI."""
import ctypes

import numpy as np
import pytest

from oracle import standin        # noqa: E402  (the tests' stand-in cosmology)

pytestmark = pytest.mark.gpu


def _solve(eng, cov, nm):
    from fastbox_amd import _lib
    N = eng.N
    cov_dev = eng.upload_raw(np.ascontiguousarray(cov, dtype=np.float64))
    modes = eng._alloc_bytes(max(1, nm) * N * 8)
    vals = eng._alloc_bytes(max(1, nm) * 8)
    sweeps = ctypes.c_int(-1)
    _lib.call("fb_leading_eigenvectors", eng._plan, cov_dev.ptr, nm, modes.ptr, vals.ptr, ctypes.byref(sweeps), eng.stream)
    U, w = np.empty((N, nm)), np.empty(nm)
    if nm:
        _lib.call("fb_memcpy_d2h", U.ctypes.data_as(ctypes.c_void_p), modes.ptr, U.nbytes, eng.stream)
        _lib.call("fb_memcpy_d2h", w.ctypes.data_as(ctypes.c_void_p), vals.ptr, w.nbytes, eng.stream)
    return U, w, sweeps.value


def _engine(N):
    from fastbox_amd import CosmoBox
    return CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=1e3, nsamp=N, realise_now=False, precision="f64").engine


@pytest.mark.parametrize("N", [16, 64, 256, 512])
def test_leading_eigenpairs_of_a_foreground_like_covariance(N):
    """Smooth power-law foregrounds + noise: eigenvalues over fourteen decades.  Eigenvalues to 1e-12 of the largest AND
    the three leading ones to 1e-7 of their own size (LAPACK itself promises an eigenvalue to 1e-16 of the LARGEST only:
    the fourth, ten decades down, it has to 2e-7); residual, orthonormality, the projector."""
    eng = _engine(N)
    rs = np.random.RandomState(N)
    nu = np.linspace(1., 2., N)
    comps = np.stack([nu ** -2.7, nu ** -2.1 * np.log(nu + 1.), nu ** -3.2, np.cos(3. * nu)], axis=1) * [1e3, 30., 3., 0.3]
    x = rs.normal(size=(4 * N, 4)) @ comps.T + 1e-4 * rs.normal(size=(4 * N, N))
    cov = np.cov(x.T)
    cov = 0.5 * (cov + cov.T)
    w_ref, v_ref = np.linalg.eigh(cov)
    w_ref, v_ref = w_ref[::-1], v_ref[:, ::-1]
    for nm in (1, 3, 7, N):
        U, w, sweeps = _solve(eng, cov, nm)
        assert 1 <= sweeps <= 20
        assert np.max(np.abs(w - w_ref[:nm])) <= 1e-12 * w_ref[0]
        assert np.all(np.abs(w[:3] - w_ref[:nm][:3]) <= 1e-7 * w_ref[:nm][:3])
        assert np.all(np.diff(w) <= 0.)
        assert np.max(np.abs(U.T @ U - np.eye(nm))) < 1e-12
        assert np.max(np.abs(cov @ U - U * w)) <= 1e-12 * w_ref[0]
        # the leading three modes one by one (LAPACK's own vectors are good to 1e-16 lambda_max / gap: 2e-9 for the third,
        # 3e-6 for the fourth)
        k = min(nm, 3)
        sgn = np.sign(np.sum(U[:, :k] * v_ref[:, :k], axis=0))
        assert np.max(np.abs(U[:, :k] * sgn - v_ref[:, :k])) < 1e-8
        if nm == 3:
            assert np.max(np.abs(U @ U.T - v_ref[:, :3] @ v_ref[:, :3].T)) < 1e-8


def test_diagonal_degenerate_and_zero_matrices():
    N = 32
    eng = _engine(N)
    d = np.arange(N, dtype=np.float64)
    d[5] = 100.; d[9] = 100.                                      # a tie: the lower index comes first
    U, w, sweeps = _solve(eng, np.diag(d), 4)
    assert sweeps == 0                                            # nothing to rotate
    assert np.array_equal(w, [100., 100., 31., 30.])
    want = np.zeros((N, 4)); want[5, 0] = want[9, 1] = want[31, 2] = want[30, 3] = 1.
    assert np.array_equal(U, want)
    U, w, sweeps = _solve(eng, np.zeros((N, N)), 3)
    assert sweeps == 0 and np.array_equal(w, np.zeros(3)) and np.array_equal(U, np.eye(N)[:, :3])
    # rank one: every direction orthogonal to v is an eigenvector of the eigenvalue 0
    v = np.random.RandomState(1).normal(size=N)
    U, w, sweeps = _solve(eng, np.outer(v, v), 2)
    assert abs(w[0] - v @ v) < 1e-12 * (v @ v) and abs(w[1]) < 1e-12 * (v @ v)
    assert np.max(np.abs(np.abs(U[:, 0]) - np.abs(v) / np.linalg.norm(v))) < 1e-13
    # nmodes = 0 is a no-op, out-of-range and non-finite input are refused
    from fastbox_amd._lib import FastBoxError
    _solve(eng, np.eye(N), 0)
    for bad in (-1, N + 1):
        with pytest.raises(FastBoxError):
            _solve(eng, np.eye(N), bad)
    c = np.eye(N); c[3, 4] = c[4, 3] = np.nan
    with pytest.raises(FastBoxError, match="non-finite"):
        _solve(eng, c, 2)


def test_the_symmetric_part_is_what_is_decomposed():
    N = 16
    eng = _engine(N)
    rs = np.random.RandomState(2)
    m = rs.normal(size=(N, N))
    U, w, _ = _solve(eng, m, N)
    w_ref = np.linalg.eigvalsh(0.5 * (m + m.T))[::-1]
    assert np.max(np.abs(w - w_ref)) < 1e-13 * np.max(np.abs(w_ref))


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_pca_filter_device_and_host_eigensolvers_agree(precision):
    """The cleaned cube depends on the span of the modes only: both solvers give the same cube, projector and
    (up to a sign per mode) amplitudes."""
    from fastbox_amd import CosmoBox, filters
    N = 64
    box = CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=1e3, nsamp=N, realise_now=False, precision=precision)
    rs = np.random.RandomState(5)
    nu = np.linspace(1., 2., N)
    data = (40. * nu ** -2.7) * (1. + 0.2 * rs.normal(size=(N, N, 1))) + (3. * nu ** -2.0) * rs.normal(size=(N, N, 1)) \
        + 0.02 * rs.normal(size=(N, N, N))
    cube = box.engine.upload(data, "real")
    for nm in (0, 1, 3, N):
        a, Ua, amps_a = filters.pca_filter(cube, nm, return_filter=True, eigensolver="device")
        b, Ub, amps_b = filters.pca_filter(cube, nm, return_filter=True, eigensolver="host")
        assert Ua.shape == Ub.shape == (N, nm) and amps_a.shape == amps_b.shape == (nm, N * N)
        assert np.max(np.abs(np.asarray(a) - np.asarray(b))) < 1e-9 * np.max(np.abs(data))
        if 0 < nm < N:
            assert np.max(np.abs(Ua @ Ua.T - Ub @ Ub.T)) < 1e-6
            k = min(nm, 2)                                  # the two foreground modes (the others sit in the noise floor)
            sgn = np.sign(np.sum(Ua[:, :k] * Ub[:, :k], axis=0))
            assert np.max(np.abs(amps_a[:k] * sgn[:, None] - amps_b[:k])) < 1e-8 * np.max(np.abs(amps_b[:k]))
    with pytest.raises(ValueError):
        filters.pca_filter(cube, 2, eigensolver="magma")
    with pytest.raises(ValueError):
        filters.pca_filter(cube, N + 1)
