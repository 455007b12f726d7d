"""Tuning aid (GPU): time the PCA cleaning steps on an N^3 foreground-dominated cube.  python tools/pca_bench.py [N]"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, ForegroundModel, NoiseModel, filters, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
box = CosmoBox(cosmo=default_cosmo, box_scale=(4e3, 4e3, 4e3), nsamp=N, redshift=0.8, realise_now=False,
               precision="f32", rng="device", seed=10)
eng = box.engine
fg = ForegroundModel(box)
cube = fg.construct_cube(fg.realise_foreground_amp(57., 1.1, 10., 4.), fg.realise_spectral_index(2.07, 0.0002, 15.)) \
    + NoiseModel(box).realise_radiometer_noise(18., 2., 1., 64) + 0.1 * box.realise_density()
cube.ptr
for rep in range(3):
    eng.sync(); t0 = time.perf_counter()
    mean = filters._channel_means(eng, cube); eng.sync(); t1 = time.perf_counter()
    cov_dev = eng._alloc_bytes(N * N * 8)
    _lib.call("fb_channel_covariance", eng._plan, cube.ptr, mean.ptr, cov_dev.ptr, eng.stream); eng.sync(); t2 = time.perf_counter()
    cov = np.empty((N, N))
    _lib.call("fb_memcpy_d2h", cov.ctypes.data_as(ctypes.c_void_p), cov_dev.ptr, cov.nbytes, eng.stream); t3 = time.perf_counter()
    w, v = np.linalg.eigh(cov); t4 = time.perf_counter()
    import scipy.linalg
    w2, v2 = scipy.linalg.eigh(cov, subset_by_index=[N - 4, N - 1]); t5 = time.perf_counter()
    out = filters.pca_filter(cube, 4); eng.sync(); t6 = time.perf_counter()
print("N=%d  means %.2f ms | covariance %.2f ms (%.1f TFLOP/s of the %d GFLOP upper triangle) | D2H %.2f ms | "
      "numpy eigh %.1f ms | scipy eigh top-4 %.1f ms | whole pca_filter %.1f ms"
      % (N, (t1 - t0) * 1e3, (t2 - t1) * 1e3, float(N) ** 4 * (1 + 64. / N) / (t2 - t1) / 1e12, float(N) ** 4 * (1 + 64. / N) / 1e9,
         (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3))
print("leading eigenvalues:", w[::-1][:6])
