"""Beam convolution on the device: the step after the density-field path that smooths every frequency channel of a
cube with the instrument beam (``fastbox/beams.py``).  Same class and method names as the reference's ``BeamModel``
(:13-137); the cubes stay in HBM.

``convolve_fft`` (:63-87) is ``scipy.signal.fftconvolve(beam, field, mode='same', axes=[0, 1])``: a LINEAR
convolution, so both cubes are zero-padded to 2N x 2N per channel and transformed with the strided FFT passes of a
2N plan (``fb_beam_convolve``, periodic = 0).  ``convolve_real`` (:90-137) is ``convolve2d(..., boundary='wrap')``
channel by channel, i.e. the CIRCULAR convolution, which the N-point transform gives directly (periodic = 1; the
reference's direct sum is O(N^4) per channel).  Both divide by the beam's sum per channel (:80, :112).

Subclasses supply ``beam_cube`` (a host ndarray or a device cube of shape (N, N, N)); the reference's own
subclasses need ``katbeam`` (absent here) or are not runnable as published (``ZernikeBeamModel.beam_cube`` refers to
an undefined ``pol``), so only the base class is mirrored.
"""
import numpy as np

from . import _lib
from .device import REAL, DeviceArray, Engine


class BeamModel(object):

    def __init__(self, box):
        """box: the CosmoBox whose grid the beam is defined on (beams.py:15-23)."""
        self.box = box

    def beam_cube(self, pol=None):
        """Beam value at every voxel of the box (beams.py:26-38: unity for the base class)."""
        return np.ones((self.box.N, self.box.N, self.box.N))

    def beam_value(self, x, y, freq, pol=None):
        """Beam value at coordinates (x, y in degrees, freq in MHz) of equal shape (beams.py:41-60)."""
        assert x.shape == y.shape == freq.shape, \
            "x, y, and freq arrays should have the same shape"
        return 1. + 0. * x

    # -- device side ------------------------------------------------------------------------------------------
    def _transform_engine(self, M):
        """Plan of the transverse transform size M on the box's device and stream (geometry tables unused)."""
        eng = self.box.engine
        if M == eng.N:
            return eng
        cache = eng.__dict__.setdefault("_transform_engines", {})
        if M not in cache:
            zeros3, zeros1 = np.zeros(3 * M), np.zeros(M)
            cache[M] = Engine(M, (1., 1., 1.), zeros3, zeros3, zeros1, np.arange(M, dtype=np.float64),
                              precision=eng.precision, device=getattr(eng, "device", 0),
                              stream=(eng.stream.value if eng.stream else None))
        return cache[M]

    def _convolve(self, field_x, pol, periodic):
        box = self.box
        eng = box.engine
        n = eng.N
        field = field_x if isinstance(field_x, DeviceArray) else box._as_real(field_x)
        if field.kind != REAL:
            raise TypeError("expected a real-space cube")
        beam = self.beam_cube(pol=pol) if pol is not None else self.beam_cube()
        M = n if periodic else 2 * n
        big = self._transform_engine(M)
        nbytes = M * M * n * np.dtype(eng.cdtype).itemsize
        work_a = big._alloc_bytes(nbytes)
        # the transform of a beam cube is kept with the cube it came from: the same array object (host or device) on
        # the next call skips its embedding and forward transform
        cached = self.__dict__.get("_beam_k")
        ready = cached is not None and cached[0] is beam and cached[1] == (M, periodic)
        if ready:
            work_b = cached[2]
            beam_ptr = None
        else:
            self._beam_k = None
            work_b = big._alloc_bytes(nbytes)
            beam_dev = beam if isinstance(beam, DeviceArray) else box._as_real(beam)
            beam_ptr = beam_dev.ptr
        out = eng.empty(REAL)
        _lib.call("fb_beam_convolve", big._plan, field.ptr, beam_ptr, work_a.ptr, work_b.ptr, out.ptr,
                  1 if periodic else 0, 1 if ready else 0, eng.stream)
        self._beam_k = (beam, (M, periodic), work_b)
        return out

    def convolve_fft(self, field_x, pol=None):
        """Beam-convolved field, every frequency channel (last axis) separately; zero-padded FFT convolution
        normalised by the beam's sum per channel (beams.py:63-87).  Returns a device cube."""
        return self._convolve(field_x, pol, periodic=False)

    def convolve_real(self, field_x, pol=None, verbose=False):
        """The reference's direct convolution with wrapped boundaries (beams.py:90-137), evaluated as the circular
        convolution it is.  Returns a device cube."""
        return self._convolve(field_x, pol, periodic=True)
