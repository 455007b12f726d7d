"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

numpy restatement of the beam-convolution step: fastbox/beams.py BeamModel.convolve_fft (:63-87)
and convolve_real (:90-137).  The reference delegates to scipy.signal (fftconvolve / convolve2d, scipy is present
here and on the GPU box); this file restates what those calls compute with explicit sums / transforms, so that the
device result is checked against the definition and not against another FFT of the same shape.  Pinned against the
reference itself by oracle/make_golden_beams.py -> tests/golden/beam_*.npz (tests/test_oracle.py).
"""
import numpy as np


def test_beam_cube(ang_x, ang_y, freqs, fwhm_deg=1.2, freq_ref=1000.):
    """A frequency-dependent, slightly elliptical and off-centre Gaussian beam on the box grid: an INPUT of the
    fixtures (the reference's base class only has the uniform beam)."""
    x, y, nu = np.meshgrid(ang_x, ang_y, freqs, indexing="ij")
    sig = fwhm_deg / 2.3548 * (freq_ref / nu)
    return np.exp(-0.5 * (((x - 0.1) / sig) ** 2. + ((y + 0.05) / (1.2 * sig)) ** 2.)) * (1. + 0.1 * np.sin(3. * x))


def convolve_fft(beam, field):
    """beams.py:79-87: linear convolution over axes 0, 1, mode='same' (centred on the first argument), divided by the
    beam's sum per channel.  Zero-padded transform of length 2N (>= 2N - 1, so nothing wraps)."""
    N = beam.shape[0]
    norm = np.sum(beam.reshape(-1, beam.shape[-1]), axis=0)
    fb = np.fft.fftn(beam, s=(2 * N, 2 * N), axes=(0, 1))
    ff = np.fft.fftn(field, s=(2 * N, 2 * N), axes=(0, 1))
    full = np.fft.ifftn(fb * ff, axes=(0, 1)).real
    o = (N - 1) // 2                                     # scipy _centered: start = (full - N) // 2, full = 2N - 1
    return full[o:o + N, o:o + N, :] / norm[np.newaxis, np.newaxis, :]


def convolve_real_direct(beam, field):
    """beams.py:129-137 by its definition (small N only): convolve2d(beam, field, mode='same', boundary='wrap') is
    out[i, j] = sum_pq field[p, q] beam[(i + o - p) mod N, (j + o - q) mod N], o = (N - 1) // 2."""
    N = beam.shape[0]
    norm = np.sum(beam.reshape(-1, beam.shape[-1]), axis=0)
    o = (N - 1) // 2
    out = np.zeros_like(field, dtype=np.float64)
    idx = np.arange(N)
    for p in range(N):
        bi = beam[(idx + o - p) % N]                     # [i][.][z]
        for q in range(N):
            out += field[p, q][np.newaxis, np.newaxis, :] * bi[:, (idx + o - q) % N, :]
    return out / norm[np.newaxis, np.newaxis, :]


def convolve_real(beam, field):
    """The same circular convolution through N-point transforms (any N)."""
    N = beam.shape[0]
    norm = np.sum(beam.reshape(-1, beam.shape[-1]), axis=0)
    circ = np.fft.ifftn(np.fft.fftn(beam, axes=(0, 1)) * np.fft.fftn(field, axes=(0, 1)), axes=(0, 1)).real
    o = (N - 1) // 2
    return np.roll(circ, (-o, -o), axis=(0, 1)) / norm[np.newaxis, np.newaxis, :]
