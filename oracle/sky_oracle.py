"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

numpy restatement of the two steps that follow the density-field path in every end-to-end flow of the
reference: radiometer noise (fastbox/noise.py:25-75) and the Gaussian foreground model
(fastbox/foregrounds.py:48-175).  Pinned against the reference itself by oracle/make_golden_sky.py ->
tests/golden/sky_*.npz (tests/test_oracle.py).
"""
import numpy as np
import scipy.ndimage

from .box_oracle import mode_numbers


def radiometer_sigma(freqs, ang_x, Tinst, tp, fov, Ndish):
    """noise.py:53-69: rms per frequency channel in mK (freqs in MHz, ang_x in degrees, tp in hours)."""
    dnu = np.abs(freqs[1] - freqs[0])
    tp = tp * 3600.
    dtheta = ang_x[1] - ang_x[0]
    t_res = tp * dtheta ** 2. / fov
    Tsky = 60e3 * (freqs / 300.) ** (-2.5)
    Tsys = Tinst * 1e3 + Tsky
    return Tsys / np.sqrt(Ndish * t_res * (dnu * 1e6))


def radiometer_noise(shape, sigma_rms, rng=np.random):
    """noise.py:72-74: unit white noise drawn in C order, scaled along the last (frequency) axis."""
    noise = rng.normal(0., 1., shape)
    noise *= sigma_rms[np.newaxis, np.newaxis, :]
    return noise


def foreground_cell(g, r, amp, beta):
    """foregrounds.py:83-97: C_ell on the 2-D (k_x, k_y) grid, ell ~ k_perp r / 2, normalised for the 2-D DFT."""
    N = g['N']
    m = mode_numbers(N)          # the numbering of box.Kx, Ky (box.py:116-123; fftfreq's for a power of two)
    # the reference slices its (N,N,N) mode cubes, Kx[:,:,0]: numpy's vectorised pow/sqrt round some elements
    # differently for strided and contiguous operands, so the operand layout is part of the restatement
    Kx = np.empty((N, N, N)); Ky = np.empty((N, N, N))
    Kx[:] = m[:, None, None]; Ky[:] = m[None, :, None]
    k_perp = 2. * np.pi * np.sqrt((Kx[:, :, 0] / g['Lx']) ** 2. + (Ky[:, :, 0] / g['Ly']) ** 2.)
    with np.errstate(all="ignore"):
        C_ell = amp * (0.5 * k_perp * r / 1000.) ** (beta)
    C_ell[np.isinf(C_ell)] = 0.
    C_ell = C_ell * ((N ** 4.) / (g['Lx'] * g['Ly']))        # one factor, as the reference's in-place *=
    return k_perp, C_ell


def foreground_amp(g, r, amp, beta, monopole, sigma_pix=None, rng=np.random):
    """foregrounds.py:99-114.  sigma_pix = smoothing_scale / pixel size in degrees (None: no smoothing)."""
    k_perp, C_ell = foreground_cell(g, r, amp, beta)
    re = rng.normal(0.0, 1.0, k_perp.shape)
    im = rng.normal(0.0, 1.0, k_perp.shape)
    fg_k = (re + 1.j * im) * np.sqrt(C_ell)
    fg_k[k_perp == 0.] = 0.
    fg_x = np.fft.ifftn(fg_k).real + monopole
    if sigma_pix is not None:
        fg_x = scipy.ndimage.gaussian_filter(fg_x, sigma=sigma_pix, mode='wrap')
    return fg_x


def spectral_index(N, mean, std, sigma_pix, rng=np.random):
    """foregrounds.py:136-145."""
    alpha = rng.normal(mean, std, (N, N))
    return scipy.ndimage.gaussian_filter(alpha, sigma=sigma_pix, mode='wrap')


def construct_cube(amps, spectral_idx, freqs, freq_ref=130.):
    """foregrounds.py:165-175."""
    if isinstance(spectral_idx, float):
        ffac = ((freqs / freq_ref) ** spectral_idx)[np.newaxis, np.newaxis, :]
    else:
        ffac = (freqs / freq_ref)[np.newaxis, np.newaxis, :] ** spectral_idx[:, :, np.newaxis]
    return amps[:, :, np.newaxis] * ffac


def gaussian_weights(sigma, truncate=4.0):
    """The 1-D kernel scipy.ndimage.gaussian_filter correlates with along each axis (order 0)."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return phi / phi.sum(), radius
