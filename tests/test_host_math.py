"""CPU: host-side models of device arithmetic whose exactness the design leans on."""
import numpy as np


def test_device_exponential_formula_is_two_ulp():
    """fb_exp (csrc/fb_fft_kernels.h): exp(x) = 2^t (1 + e ln 2), t = fl(x c), e = fma(x, c, -t) + x (log2 e - c),
    c = float(log2 e).  Emulated in float32 with exact fused multiply-adds and a correctly rounded 2^t: at most 2 ulp
    from exp(x) over the range a shifted log-normal field can reach (the plain 2^(x log2 e) form is 60 ulp off at |x| = 80)."""
    x = np.linspace(-80, 60, 400001).astype(np.float32)
    c = np.float32(1.44269502162933349609375)
    x64 = x.astype(np.float64)
    t = (x * c).astype(np.float32)
    e = (x64 * np.float64(c) - t.astype(np.float64)).astype(np.float32)
    e = (x64 * 1.925963033500011e-8 + e.astype(np.float64)).astype(np.float32)
    r = np.exp2(t.astype(np.float64)).astype(np.float32)
    res = (r.astype(np.float64) * (np.float64(np.float32(0.693147182464599609375)) * e.astype(np.float64))
           + r.astype(np.float64)).astype(np.float32)
    true = np.exp(x64)
    ulp = np.abs(res.astype(np.float64) / true - 1) / 2.0 ** -24
    assert ulp.max() < 2.0
    naive = np.exp2((x * np.float32(1.4426950408889634)).astype(np.float32).astype(np.float64))
    assert (np.abs(naive / true - 1) / 2.0 ** -24).max() > 30.0
    assert abs(float(c) + 1.925963033500011e-8 - np.log2(np.e)) < 1e-15
