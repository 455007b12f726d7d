"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

numpy restatement of the foreground-cleaning step of the reference's end-to-end flow:
fastbox/filters.py mean_spectrum_filter (:35-56) and pca_filter (:93-183, both settings of fit_powerlaw).  Pinned against the
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


def pca_filter(field, nmodes, fit_powerlaw=False, return_filter=False):
    """filters.py:139-183.  fit_powerlaw: x = d - (power-law fit of the mean spectrum), :146-154; np.cov(x) centres
    every channel on its own mean again, so the covariance is unchanged by the choice of what is subtracted."""
    d_mean, x, cov = channel_covariance(field)
    if fit_powerlaw:
        from scipy.optimize import curve_fit
        d = field.reshape((-1, field.shape[-1])).T
        freqs = np.linspace(1., 10., d.shape[0])

        def fn(nu, amp, beta):
            return amp * (nu / nu[0]) ** beta
        pfit, _ = curve_fit(fn, freqs, d_mean.flatten(), p0=[d_mean[0][0], -2.7])
        d_mean = fn(freqs, pfit[0], pfit[1])[:, np.newaxis]
        x = d - d_mean
        cov = np.cov(x)
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
