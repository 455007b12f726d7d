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


def test_device_exponential_overflows_to_infinity_not_nan():
    """fb_exp past the float range: 2^t is infinite and, where the first-order correction is exactly 0, inf * 0 + inf
    would be NaN; the device returns the infinity (as libm's expf does), which the bin sums' range check then sees."""
    x = np.array([88.0, 88.72284, 89.0, 128.0 / 1.44269502162933349609375, 200.0], dtype=np.float32)
    c = np.float32(1.44269502162933349609375)
    t = (x * c).astype(np.float32)
    e = (x.astype(np.float64) * np.float64(c) - t.astype(np.float64)).astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        r = np.exp2(t.astype(np.float64)).astype(np.float32)
        naive = (r.astype(np.float64) * (0.693147182464599609375 * e.astype(np.float64)) + r.astype(np.float64)).astype(np.float32)
        guarded = np.where(r > np.float32(3.4028234663852886e38), r, naive)
    assert np.isfinite(guarded[0]) and np.all(np.isinf(guarded[2:])) and not np.any(np.isnan(guarded))


def test_lognormal_shift_keeps_the_sum_of_exponentials_in_single_precision_range():
    """hostgeom.lognormal_shift: for Gaussian samples of any sigma the shifted sum S = sum exp(d - c) lands where a
    single-precision plan can carry it -- S^4 (the k = 0 mode's |.|^4) below 2^128, and S^4 times the 1e-24 of the
    smallest |delta_k|^4 / S^4 ratio a mean-dominated field has above 2^-126 -- and round 2's sigma^2/2 does not."""
    from fastbox_amd import hostgeom
    rs = np.random.RandomState(3)
    n = 1 << 22
    g = rs.standard_normal(n)
    for sigma in (0.3, 1.0, 3.0, 5.5, 8.3, 14.0, 21.4, 40.0):
        d = sigma * g
        c = hostgeom.lognormal_shift(sigma * sigma, n)
        lnS = np.log(np.sum(np.exp(d - d.max()))) + d.max() - c
        assert -12.0 < lnS < hostgeom.LN_SUM_MAX, (sigma, lnS)
        if sigma * sigma <= 2 * np.log(n):
            assert -6.0 < lnS                                   # mean-dominated: small modes must not underflow either
        exact = hostgeom.lognormal_shift_exact(d.max(), n)
        lnS2 = np.log(np.sum(np.exp(d - d.max()))) + d.max() - exact
        assert -2.0 <= lnS2 <= hostgeom.LN_SUM_MAX
    # the bench's 2048^3 box: sigma = 21.4, 8.6e9 voxels -- sigma^2/2 puts the LARGEST exponent at -94
    sigma, nv = 21.4, 2048.0 ** 3
    a = np.sqrt(2 * np.log(nv))
    top = sigma * (a - (np.log(np.log(nv)) + np.log(4 * np.pi)) / (2 * a))        # expected maximum: 6.3 sigma
    assert top - 0.5 * sigma ** 2 < -87.3                                           # below the smallest normal float
    assert -40.0 < top - hostgeom.lognormal_shift(sigma ** 2, nv) < 10.0
    assert hostgeom.lognormal_shift(0.0, nv) == 0.0


def test_lognormal_range_check():
    from fastbox_amd import hostgeom
    cnt = np.array([1., 6., 0., 24.])
    s1 = np.array([5., 12., 0., 48.])
    s2 = np.array([25., 24., 0., 100.])
    assert hostgeom.lognormal_sums_in_range(cnt, s1, s2, 10.0)
    assert not hostgeom.lognormal_sums_in_range(cnt, s1, s2, 0.0)
    assert not hostgeom.lognormal_sums_in_range(cnt, s1, s2, np.inf)
    assert not hostgeom.lognormal_sums_in_range(cnt, s1 * np.array([1, np.inf, 1, 1]), s2, 10.0)
    assert not hostgeom.lognormal_sums_in_range(cnt, s1, s2 * np.array([1, 1, 1, 0.5]), 10.0)   # squares flushed to zero
    assert hostgeom.lognormal_sums_in_range(cnt, s1 * np.array([np.inf, 1, 1, 1]), s2, 10.0)     # bin 0 is discarded


def test_batched_spectrum_arithmetic_equals_the_scalar_path():
    """hostgeom.finish_bins_many / lognormal_sums_in_range_many (what a batch of fetched records goes through) give the
    numbers of the per-record functions bit for bit, NaN masks and out-of-range verdicts included."""
    from fastbox_amd import hostgeom
    rng = np.random.RandomState(3)
    nb, m = 20, 37
    cnt = np.array([0., 1., 2., 0.] + list(rng.randint(3, 5000, nb - 4)), dtype=np.float64)
    s1 = rng.rand(m, nb) * cnt * 3.
    s2 = (s1 / np.maximum(cnt, 1)) ** 2 * cnt * (1. + rng.rand(m, nb))
    s1[:, cnt == 0] = 0.
    s2[:, cnt == 0] = 0.
    s2[:, 2] = s1[:, 2] ** 2 / 2. * (1. + 1e-8)          # a two-mode bin with (nearly) zero spread
    esum = rng.rand(m) * 1e6 + 1.
    s1[5, 7] = np.inf; s2[9, 6] *= 1e-3; esum[11] = 0.; s1[13, 8] = 0.; s2[13, 8] = 0.; s1[14, 9] = 0.
    many_ok = hostgeom.lognormal_sums_in_range_many(cnt, s1, s2, esum)
    vals, std = hostgeom.finish_bins_many(cnt, s1, s2, 0.37, 2. ** -23)
    for i in range(m):
        assert many_ok[i] == hostgeom.lognormal_sums_in_range(cnt, s1[i], s2[i], esum[i]), i
        v, s = hostgeom.finish_bins(cnt, s1[i], s2[i], 0.37, 2. ** -23)
        assert np.array_equal(v, vals[i], equal_nan=True) and np.array_equal(s, std[i], equal_nan=True), i
    assert not many_ok[5] and not many_ok[9] and not many_ok[11] and many_ok[13] and not many_ok[14] and many_ok[0]
