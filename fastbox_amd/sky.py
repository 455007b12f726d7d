"""
The two steps that follow the density-field path in the reference's end-to-end flow
(examples/example_endtoend.py:58-87), on the device: the Gaussian foreground model
(fastbox/foregrounds.py:34-175) and radiometer noise (fastbox/noise.py:11-75).  Same class and
method names, arguments and return shapes as the reference; 2-D maps come back as float64
ndarrays (N, N) like the reference's, cubes as ``DeviceArray`` (they stay in HBM and add to
the box's fields there).  No CPU fallback: every array operation runs through libfastbox_hip.

Random numbers follow the box: ``rng='numpy'`` draws the reference's legacy global stream on the
host in the reference's order (same seed => same maps / cube), ``rng='device'`` uses the counter
generator (streams 2-4).
"""
import ctypes

import numpy as np

from . import _lib
from . import box as _box
from .device import REAL, DeviceArray

_ccl = _box._ccl


def _gaussian_weights(sigma, truncate=4.0):
    """The 1-D kernel scipy.ndimage.gaussian_filter correlates with (order 0, scipy/ndimage/_filters.py)."""
    sd = float(sigma)
    radius = int(truncate * sd + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sd * sd) * x ** 2)
    return np.ascontiguousarray(phi / phi.sum()), radius


class _Maps(object):
    """Device helpers shared by the two models: N x N maps in the plan's precision."""

    def __init__(self, box):
        self.box = box
        self.eng = box.engine
        self.N = box.N

    def upload(self, a, complex_=False):
        dt = self.eng.cdtype if complex_ else self.eng.rdtype
        return self.eng.upload_raw(np.ascontiguousarray(a, dtype=dt))

    def empty(self, complex_=False):
        item = np.dtype(self.eng.cdtype if complex_ else self.eng.rdtype).itemsize
        return self.eng._alloc_bytes(self.N * self.N * item)

    def download(self, buf):
        h = np.empty((self.N, self.N), dtype=self.eng.rdtype)
        _lib.call("fb_memcpy_d2h", h.ctypes.data_as(ctypes.c_void_p), buf.ptr, h.nbytes, self.eng.stream)
        return h.astype(np.float64)

    def smooth(self, buf, sigma_pix):
        w, radius = _gaussian_weights(sigma_pix)
        tmp = self.empty()
        _lib.call("fb_sky_gaussian_filter", self.eng._plan, buf.ptr, tmp.ptr, w.ctypes.data_as(_lib.P_double),
                  int(radius), self.eng.stream)
        self.eng.sync()                      # `w` must outlive the copy
        return buf

    def next_seed(self):
        b = self.box
        b._sky_draws = getattr(b, "_sky_draws", 0) + 1
        return (b.seed * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03 * b._sky_draws) & (2 ** 64 - 1)


class ForegroundModel(object):

    def __init__(self, box):
        """Foregrounds on top of a box (foregrounds.py:36-46)."""
        self.box = box
        self._m = _Maps(box)

    def realise_foreground_amp(self, amp, beta, monopole, smoothing_scale=None, redshift=None):
        """2-D Gaussian random field with C_ell = amp (ell/1000)^beta, ell ~ k_perp r / 2, plus a monopole,
        optionally smoothed (foregrounds.py:48-114).  Returns an (N, N) float64 array."""
        box, m = self.box, self._m
        if redshift is None:
            redshift = box.redshift
        scale_factor = 1. / (1. + redshift)
        r = _ccl.comoving_angular_distance(box.cosmo, scale_factor)
        mo = box._modes
        # foregrounds.py:83-97, on the (k_x, k_y) grid: N^2 values, evaluated once on the host like P(k)
        k_perp = 2. * np.pi * np.sqrt(((mo / box.Lx) ** 2.)[:, None] + ((mo / box.Ly) ** 2.)[None, :])
        with np.errstate(all="ignore"):
            C_ell = amp * (0.5 * k_perp * r / 1000.) ** (beta)
        C_ell[np.isinf(C_ell)] = 0.
        C_ell = C_ell * ((box.N ** 4.) / (box.Lx * box.Ly))
        amp2d = np.sqrt(C_ell)
        amp2d[k_perp == 0.] = 0.                                         # :103
        if box.rng == "numpy":
            re = m.upload(np.random.normal(0.0, 1.0, k_perp.shape))
            im = m.upload(np.random.normal(0.0, 1.0, k_perp.shape))
            re_p, im_p, seed = re.ptr, im.ptr, 0
        else:
            re_p, im_p, seed = None, None, m.next_seed()
        out, work = m.empty(), m.empty(complex_=True)
        _lib.call("fb_sky_realise_map", m.eng._plan, m.upload(amp2d).ptr, re_p, im_p, seed, float(monopole),
                  work.ptr, out.ptr, m.eng.stream)
        if smoothing_scale is not None:
            ang_x, ang_y = box.pixel_array(redshift=redshift)
            m.smooth(out, smoothing_scale / (ang_x[1] - ang_x[0]))
        return m.download(out)

    def realise_spectral_index(self, mean_spec_idx, std_spec_idx, smoothing_scale, redshift=None):
        """Gaussian spectral-index map, smoothed (foregrounds.py:116-145).  Returns (N, N) float64."""
        box, m = self.box, self._m
        out = m.empty()
        if box.rng == "numpy":
            # normal(mean, std) of the legacy stream = mean + std * gauss(): draw it exactly so, then upload
            alpha = np.random.normal(mean_spec_idx, std_spec_idx, (box.N, box.N))
            unit = None
            buf = m.upload(alpha)
            _lib.call("fb_memcpy_d2d", out.ptr, buf.ptr, out.nbytes, m.eng.stream)
        else:
            _lib.call("fb_sky_normal_map", m.eng._plan, None, m.next_seed(), float(mean_spec_idx),
                      float(std_spec_idx), out.ptr, m.eng.stream)
        ang_x, ang_y = box.pixel_array(redshift=redshift)
        m.smooth(out, smoothing_scale / (ang_x[1] - ang_x[0]))
        return m.download(out)

    def construct_cube(self, amps, spectral_idx, freq_ref=130., redshift=None):
        """amps[x, y] * (freqs / freq_ref) ** spectral_idx[x, y] as a device cube (foregrounds.py:147-175)."""
        box, m = self.box, self._m
        freqs = box.freq_array(redshift=redshift)
        ratio = np.ascontiguousarray(freqs / freq_ref, dtype=np.float64)
        a = m.upload(np.asarray(amps, dtype=np.float64))
        if isinstance(spectral_idx, float):
            al_p, al_s = None, float(spectral_idx)
        else:
            al = m.upload(np.asarray(spectral_idx, dtype=np.float64))
            al_p, al_s = al.ptr, 0.0
        out = m.eng.empty(REAL)
        _lib.call("fb_sky_foreground_cube", m.eng._plan, a.ptr, al_p, al_s, ratio.ctypes.data_as(_lib.P_double),
                  out.ptr, m.eng.stream)
        return out


class NoiseModel(object):

    def __init__(self, box):
        """Noise on top of a box (noise.py:13-22)."""
        self.box = box
        self._m = _Maps(box)

    def realise_radiometer_noise(self, Tinst, tp, fov, Ndish, redshift=None):
        """White noise with the radiometer-equation rms per channel, in mK (noise.py:25-75), as a device cube."""
        box, m = self.box, self._m
        freqs = box.freq_array(redshift=redshift)
        dnu = np.abs(freqs[1] - freqs[0])
        tp = tp * 3600.
        ang_x, ang_y = box.pixel_array(redshift=redshift)
        dtheta = ang_x[1] - ang_x[0]
        t_res = tp * dtheta ** 2. / fov
        Tsky = 60e3 * (freqs / 300.) ** (-2.5)
        Tsys = Tinst * 1e3 + Tsky
        sigma_rms = np.ascontiguousarray(Tsys / np.sqrt(Ndish * t_res * (dnu * 1e6)), dtype=np.float64)
        out = m.eng.empty(REAL)
        if box.rng == "numpy":
            unit = m.eng.upload(np.random.normal(0., 1., (box.N, box.N, box.N)), REAL)
            unit_p, seed = unit.ptr, 0
        else:
            unit_p, seed = None, m.next_seed()
        _lib.call("fb_sky_noise_cube", m.eng._plan, sigma_rms.ctypes.data_as(_lib.P_double), unit_p, seed, out.ptr,
                  m.eng.stream)
        m.eng.sync()                         # `sigma_rms` must outlive the copy
        return out
