"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

numpy restatement of the foreground-cleaning step of the reference's end-to-end flow:
fastbox/filters.py mean_spectrum_filter (:35-56) and pca_filter (:93-183, fit_powerlaw=False).  Pinned against the
reference itself by oracle/make_golden_sky.py -> tests/golden/pca_*.npz (tests/test_oracle.py).
"""
import numpy as np


def mean_spectrum_filter(field):
    """filters.py:49-56."""
    d = field.reshape((-1, field.shape[-1]))
    d_mean = np.mean(d, axis=0)[np.newaxis, :]
    return (d - d_mean).reshape(field.shape)


def channel_covariance(field):
    """filters.py:139-158: mean spectrum and the frequency-frequency covariance np.cov forms (divisor Npix - 1)."""
    d = field.reshape((-1, field.shape[-1])).T
    d_mean = np.mean(d, axis=-1)[:, np.newaxis]
    x = d - d_mean
    return d_mean, x, np.cov(x)


def pca_filter(field, nmodes, return_filter=False):
    """filters.py:139-183 with fit_powerlaw=False."""
    d_mean, x, cov = channel_covariance(field)
    eigvals, eigvecs = np.linalg.eig(cov)
    idxs = np.argsort(eigvals)[::-1]
    eigvals = eigvals[idxs]
    eigvecs = eigvecs[:, idxs]
    U_fg = eigvecs[:, :nmodes]
    fg_amps = np.dot(U_fg.T, x)
    fg_field = np.dot(U_fg, fg_amps) + d_mean
    fg_field = fg_field.T.reshape(field.shape)
    cleaned_field = field - fg_field
    if return_filter:
        return cleaned_field, U_fg, fg_amps
    return cleaned_field


def angular_bandpass_filter(field, kmin, kmax, d=1.):
    """filters.py:79-90: top-hat band-pass in |k_perp| of every frequency channel (2-D transforms over axes 0, 1)."""
    field_k = np.fft.fftn(field, axes=[0, 1])
    kx = np.fft.fftfreq(field.shape[0], d=d)
    kx, ky = np.meshgrid(kx, kx)
    k = np.sqrt(kx ** 2. + ky ** 2.)
    field_k[~np.logical_and(k >= kmin, k < kmax)] *= 0.
    return np.fft.ifftn(field_k, axes=[0, 1])
