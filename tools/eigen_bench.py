"""Tuning aid: fb_leading_eigenvectors (Jacobi on the device) against LAPACK on the host for a foreground-like covariance.
   python tools/eigen_bench.py [N ...]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from fastbox_amd import CosmoBox, _lib, default_cosmo            # noqa: E402
from fastbox_amd.filters import _few_blas_threads                 # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 512, 1024, 2048]
    print("# tools/eigen_bench.py: 8 leading eigenpairs of an N x N covariance (power-law foregrounds + noise), ms")
    for N in sizes:
        eng = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f32", rng="device").engine
        rs = np.random.RandomState(N)
        nu = np.linspace(1., 2., N)
        comps = np.stack([nu ** -2.7, nu ** -2.1 * np.log(nu + 1.), nu ** -3.2, np.cos(3. * nu)], axis=1) * [1e3, 30., 3., 0.3]
        x = rs.normal(size=(4 * N, 4)) @ comps.T + 1e-3 * rs.normal(size=(4 * N, N))
        cov = np.cov(x.T)
        cov = 0.5 * (cov + cov.T)
        cov_dev = eng.upload_raw(cov)
        modes = eng._alloc_bytes(8 * N * 8)
        sweeps = ctypes.c_int(0)
        ts = []
        for _ in range(3):
            _lib.call("fb_stream_sync", eng.stream)
            t0 = time.perf_counter()
            _lib.call("fb_leading_eigenvectors", eng._plan, cov_dev.ptr, 8, modes.ptr, None, ctypes.byref(sweeps), eng.stream)
            ts.append(time.perf_counter() - t0)
        from scipy.linalg import eigh
        th = []
        with _few_blas_threads():
            for _ in range(3):
                t0 = time.perf_counter()
                eigh(cov, subset_by_index=[N - 8, N - 1])
                th.append(time.perf_counter() - t0)
        U = np.empty((N, 8))
        _lib.call("fb_memcpy_d2h", U.ctypes.data_as(ctypes.c_void_p), modes.ptr, U.nbytes, eng.stream)
        w, v = eigh(cov, subset_by_index=[N - 3, N - 1])
        proj = np.max(np.abs(U[:, :3] @ U[:, :3].T - v @ v.T))
        print("N %5d: device %8.1f ms (%2d sweeps, %6d launches)   host LAPACK dsyevr (2 threads) %8.1f ms   "
              "projector on the 3 leading modes differs by %.1e" % (N, 1e3 * min(ts), sweeps.value, sweeps.value * (N - 1),
                                                                   1e3 * min(th), proj))
        eng.close()


if __name__ == "__main__":
    main()
