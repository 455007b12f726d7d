"""
Foreground cleaning of a data cube on the device: the step between "add foregrounds and noise" and
"estimate the power spectrum" in the reference's end-to-end flow (examples/example_endtoend.py:105-111).
Same function names, arguments and return values as fastbox/filters.py (:35-56, :93-183).

The cube never leaves HBM: channel means, the frequency-frequency covariance (N^2 pixels x N x N
channels, on the fp64 matrix cores) and the projection run in libfastbox_hip.  The leading eigenvectors
of the N x N covariance come from LAPACK on the host by default (the faster of the two for one small matrix)
or, with `eigensolver="device"`, from fb_leading_eigenvectors (cyclic Jacobi in fp64 on the GPU: the route
of a C-ABI consumer without LAPACK).  The cleaned cube depends only on the span of the modes, so it equals the reference's
(which uses the unsymmetric solver numpy.linalg.eig); individual eigenvectors and mode amplitudes may
differ from the reference's by a sign.
"""
import ctypes

import numpy as np

from . import _lib
from .device import FULL, REAL, DeviceArray


def _few_blas_threads():
    """An N x N eigenproblem does not feed 64 BLAS threads: on a many-core GPU host the default pool makes it
    10-40x slower (5 ms with two threads, 60-200 ms with 64, measured on the MI355X boxes)."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=2)
    except Exception:
        import contextlib
        return contextlib.nullcontext()


def _as_cube(field, box=None):
    if isinstance(field, DeviceArray):
        if field.kind != REAL:
            raise TypeError("expected a real-space cube")
        return field
    if box is None:
        raise TypeError("a host array needs `box=` (the CosmoBox whose engine holds the cube)")
    return box._as_real(field)


def _channel_means(eng, cube):
    mean = eng._alloc_bytes(eng.N * 8)
    _lib.call("fb_channel_means", eng._plan, cube.ptr, mean.ptr, eng.stream)
    return mean


def mean_spectrum_filter(field, box=None):
    """Subtract the mean of every frequency channel (filters.py:35-56)."""
    cube = _as_cube(field, box)
    eng = cube.engine
    mean = _channel_means(eng, cube)
    out = eng.empty(REAL)
    _lib.call("fb_pca_clean", eng._plan, cube.ptr, mean.ptr, None, 0, out.ptr, None, eng.stream)
    return out


def pca_filter(field, nmodes, fit_powerlaw=False, return_filter=False, box=None, eigensolver="host"):
    """Remove the `nmodes` leading eigenmodes of the empirical frequency-frequency covariance
    (filters.py:93-183).  Returns the cleaned cube (DeviceArray) and, if `return_filter`, the mode matrix
    U_fg (Nfreq, nmodes) and the amplitudes fg_amps (nmodes, Npix) as host arrays.
    eigensolver: "host" (default: LAPACK dsyevr on the downloaded N x N covariance -- 6.5 ms at N = 512) or "device"
    (fb_leading_eigenvectors: cyclic Jacobi in fp64 on the GPU, nothing leaves it and no LAPACK is needed -- 30 ms at
    N = 512, profiles/r04_eigen_bench.txt).  The sign of a mode is not defined in either, as in the reference."""
    if eigensolver not in ("device", "host"):
        raise ValueError("eigensolver: 'device' or 'host'")
    cube = _as_cube(field, box)
    eng = cube.engine
    N = eng.N
    mean_true = _channel_means(eng, cube)
    mean = mean_true
    if fit_powerlaw:
        # filters.py:146-154: x = d - (power-law fit of the mean spectrum) is what the modes are projected out of and
        # what is added back; np.cov(x) (:157-158) still centres every channel on its own mean, so the covariance
        # is taken about the TRUE channel means -- the fit only enters the projection step (N numbers: host)
        from scipy.optimize import curve_fit
        h = np.empty(N)
        _lib.call("fb_memcpy_d2h", h.ctypes.data_as(ctypes.c_void_p), mean_true.ptr, h.nbytes, eng.stream)
        freqs = np.linspace(1., 10., N)

        def fn(nu, amp, beta):
            return amp * (nu / nu[0]) ** beta
        pfit, _ = curve_fit(fn, freqs, h, p0=[h[0], -2.7])
        mean = eng.upload_raw(np.ascontiguousarray(fn(freqs, pfit[0], pfit[1])))
    cov_dev = eng._alloc_bytes(N * N * 8)
    _lib.call("fb_channel_covariance", eng._plan, cube.ptr, mean_true.ptr, cov_dev.ptr, eng.stream)
    nmodes = int(nmodes)
    if not 0 <= nmodes <= N:
        raise ValueError("nmodes must lie in 0 .. %d" % N)
    # filters.py:161-169: eigenvectors by decreasing eigenvalue, keep nmodes
    U_fg = None
    if eigensolver == "device":
        U_dev = eng._alloc_bytes(max(1, nmodes) * N * 8)
        _lib.call("fb_leading_eigenvectors", eng._plan, cov_dev.ptr, nmodes, U_dev.ptr, None, None, eng.stream)
    else:
        cov = np.empty((N, N))
        _lib.call("fb_memcpy_d2h", cov.ctypes.data_as(ctypes.c_void_p), cov_dev.ptr, cov.nbytes, eng.stream)
        with _few_blas_threads():
            if 0 < nmodes < N:
                from scipy.linalg import eigh      # only the nmodes largest eigenpairs (LAPACK dsyevr)
                w, v = eigh(cov, subset_by_index=[N - nmodes, N - 1])
            else:
                w, v = np.linalg.eigh(cov)
        U_fg = np.ascontiguousarray(v[:, ::-1][:, :nmodes])
        U_dev = eng.upload_raw(U_fg)
    out = eng.empty(REAL)
    amps_dev = eng._alloc_bytes(max(1, nmodes) * N * N * 8) if return_filter else None
    _lib.call("fb_pca_clean", eng._plan, cube.ptr, mean.ptr, U_dev.ptr, int(nmodes), out.ptr,
              amps_dev.ptr if amps_dev is not None else None, eng.stream)
    if not return_filter:
        return out
    fg_amps = np.empty((nmodes, N * N))
    if nmodes:
        _lib.call("fb_memcpy_d2h", fg_amps.ctypes.data_as(ctypes.c_void_p), amps_dev.ptr, fg_amps.nbytes, eng.stream)
    if U_fg is None:
        U_fg = np.empty((N, nmodes))
        if nmodes:
            _lib.call("fb_memcpy_d2h", U_fg.ctypes.data_as(ctypes.c_void_p), U_dev.ptr, U_fg.nbytes, eng.stream)
    return out, U_fg, fg_amps


def angular_bandpass_filter(field, kmin, kmax, d=1., box=None):
    """Top-hat band-pass in |k_perp| applied to every frequency channel (filters.py:58-90).  Returns the complex
    cube ifftn(fftn(field, axes=[0, 1]) * mask, axes=[0, 1]) as a device array, like the reference's result."""
    if isinstance(field, DeviceArray) and field.kind == FULL:
        eng = field.engine
        cube = eng.clone(field)
    else:
        real = _as_cube(field, box)
        eng = real.engine
        cube = eng.empty(FULL)
        _lib.call("fb_real_to_complex", eng._plan, real.ptr, cube.ptr, eng.stream)
    N = eng.N
    kx = np.fft.fftfreq(N, d=d)                               # :83-86, N^2 numbers on the host
    kx, ky = np.meshgrid(kx, kx)
    k = np.sqrt(kx ** 2. + ky ** 2.)
    mask = np.logical_and(k >= kmin, k < kmax).astype(eng.rdtype)
    mask_dev = eng.upload_raw(np.ascontiguousarray(mask))
    _lib.call("fb_fft_transverse", eng._plan, cube.ptr, -1, eng.stream)
    _lib.call("fb_mask_transverse", eng._plan, cube.ptr, mask_dev.ptr, eng.stream)
    _lib.call("fb_fft_transverse", eng._plan, cube.ptr, +1, eng.stream)
    cube.invalidate()
    return cube
