"""
HBM-resident fields and the per-box engine that drives libfastbox_hip.

``Engine`` owns one ``fb_plan`` (grid geometry, twiddles, lookup tables) and a
small pool of device buffers; every method is a thin call into the C ABI.
``DeviceArray`` is what CosmoBox methods return: the field stays in HBM and is
only copied to the host (as float64 / complex128, the reference's dtypes) when
numpy touches it (``np.asarray``, indexing, ufuncs).
"""
import ctypes
import os
import weakref

import numpy as np
from numpy.lib.mixins import NDArrayOperatorsMixin

from . import _lib

REAL, HALF, FULL = "real", "half", "full"


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _Buffer(object):
    """Raw device allocation returned to the engine's pool when dropped."""

    __slots__ = ("ptr", "nbytes", "pool")

    def __init__(self, ptr, nbytes, pool):
        self.ptr, self.nbytes, self.pool = ptr, nbytes, pool

    def __del__(self):
        pool = self.pool
        if pool is not None and self.ptr:
            closed = pool.get("closed")
            if closed is not None:                 # the engine is gone: nothing will reuse the buffer
                try:
                    closed.fb_free(ctypes.c_void_p(self.ptr))
                except Exception:
                    pass
            else:
                pool.setdefault(self.nbytes, []).append(self.ptr)
            self.ptr = None


class _ResultSlot(object):
    """One (2*nbins+1)-double record of the engine's result ring on the device (bin sums of a queued
    power spectrum).  All records queued since the last fetch come back in a single device-to-host copy."""

    __slots__ = ("ptr", "serial")

    def __init__(self, ptr, serial):
        self.ptr, self.serial = ptr, serial


class DeviceArray(NDArrayOperatorsMixin):
    """A (N,N,N) field living in device memory.

    kind 'real': T[N][N][N]; 'half': Hermitian half spectrum (host view is the
    full (N,N,N) complex array); 'full': complex [N][N][N].
    """

    __array_priority__ = 1000

    def __init__(self, engine, kind, buf, as_complex=False):
        self.engine, self.kind, self._buf = engine, kind, buf
        self._host = None
        self._as_complex = as_complex   # real field presented as complex (apply_transfer_fn returns complex)
        N = engine.N
        self.shape = (N, N, N)
        self.ndim = 3
        self.size = N ** 3

    @property
    def ptr(self):
        return self._buf.ptr

    @property
    def dtype(self):
        return np.dtype(np.float64 if (self.kind == REAL and not self._as_complex) else np.complex128)

    def __len__(self):
        return self.shape[0]

    def host(self):
        """Host copy in the reference's dtype (cached; read-only snapshot)."""
        if self._host is None:
            h = self.engine.download(self)
            if self._as_complex:
                h = h.astype(np.complex128)
            h.setflags(write=False)
            self._host = h
        return self._host

    def invalidate(self):
        self._host = None

    def __array__(self, dtype=None, copy=None):
        h = self.host()
        if dtype is not None and np.dtype(dtype) != h.dtype:
            return h.astype(dtype)
        return h

    def _device_arith(self, ufunc, inputs):
        """+, -, * between real cubes of one engine, or a real cube and a Python scalar, without leaving the GPU
        (the arithmetic callers of the reference write between the steps of a pipeline).  None: not handled."""
        if ufunc not in (np.add, np.subtract, np.multiply, np.negative) or not all(
                (isinstance(x, DeviceArray) and x.kind == REAL and not x._as_complex and x.engine is self.engine)
                or isinstance(x, (int, float, np.floating, np.integer)) for x in inputs):
            return None
        eng = self.engine
        dev = [x for x in inputs if isinstance(x, DeviceArray)]
        if ufunc is np.negative:
            return eng.axpby(inputs[0], None, -1.0, 0.0, 0.0)
        x, y = inputs
        if len(dev) == 2:
            if ufunc is np.multiply:
                return eng.multiply(x, y)
            return eng.axpby(x, y, 1.0, 1.0 if ufunc is np.add else -1.0, 0.0)
        s, d, first = (float(y), x, True) if isinstance(x, DeviceArray) else (float(x), y, False)
        if ufunc is np.multiply:
            return eng.axpby(d, None, s, 0.0, 0.0)
        if ufunc is np.add:
            return eng.axpby(d, None, 1.0, 0.0, s)
        return eng.axpby(d, None, 1.0, 0.0, -s) if first else eng.axpby(d, None, -1.0, 0.0, s)     # d - s, s - d

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method == "__call__" and not kwargs:
            out = self._device_arith(ufunc, inputs)
            if out is not None:
                return out
        args = [x.host() if isinstance(x, DeviceArray) else x for x in inputs]
        if "out" in kwargs:
            return NotImplemented
        return getattr(ufunc, method)(*args, **kwargs)

    # `cube += other` rebinds the name to a new device cube (device fields are immutable snapshots; other
    # references to the old cube keep its values)
    def __iadd__(self, other):
        return self + other

    def __isub__(self, other):
        return self - other

    def __imul__(self, other):
        return self * other

    def __getitem__(self, idx):
        return self.host()[idx]

    @property
    def real(self):
        if self.kind == REAL:
            self.ptr                            # materialise deferred fields
            return DeviceArray(self.engine, REAL, self._buf)
        return self.host().real

    @property
    def imag(self):
        return self.host().imag

    def flatten(self):
        return self.host().flatten()

    def copy(self):
        return np.array(self.host())

    def __repr__(self):
        return "DeviceArray(kind=%s, shape=%s, precision=%s)" % (self.kind, self.shape, self.engine.precision)


def new_stream(device=None, priority=None):
    """A fresh non-blocking HIP stream (raw handle) for ``CosmoBox(..., stream=...)``: independent
    boxes on different streams overlap on the GPU (compute-bound passes of one with memory-bound
    passes of the other).  ``device``: the GPU the stream is for (default: the current one);
    ``priority``: None (default), or < 0 / 0 / > 0 for the device's highest / middle / lowest stream priority."""
    s = ctypes.c_void_p()

    def make():
        if priority is None:
            _lib.call("fb_stream_create", ctypes.byref(s))
        else:
            _lib.call("fb_stream_create_priority", ctypes.byref(s), int(priority))
    if device is None:
        make()
    else:
        with _lib.on_device(device):
            make()
    return s.value


_ENGINES = weakref.WeakSet()        # live engines (their idle-buffer pools are released when memory runs out)


class Engine(object):
    """One fb_plan + buffer pool.  All field arguments are DeviceArrays."""

    def __init__(self, N, L, axis2, ksc, kpar, zgrid, precision="f32", device=0, stream=None):
        if precision not in ("f32", "f64"):
            raise ValueError("precision must be 'f32' or 'f64'")
        self.lib = _lib.load()
        self.N = int(N)
        self.precision = precision
        self.device = int(device)
        self.rdtype = np.float32 if precision == "f32" else np.float64
        self.cdtype = np.complex64 if precision == "f32" else np.complex128
        self.stream = ctypes.c_void_p(stream) if stream else None
        self._pool = {}
        self._plan = ctypes.c_void_p()
        tabs = [np.ascontiguousarray(t, dtype=np.float64) for t in (axis2, ksc, kpar, zgrid)]
        assert tabs[0].size == 3 * N and tabs[1].size == 3 * N and tabs[2].size == N and tabs[3].size == N
        _lib.call("fb_plan_create", ctypes.byref(self._plan), self.N, float(L[0]), float(L[1]), float(L[2]),
                  4 if precision == "f32" else 8, int(device),
                  *[t.ctypes.data_as(_lib.P_double) for t in tabs])
        self.pitch = self.lib.fb_half_pitch(self._plan)
        self.rows = self.lib.fb_half_rows(self._plan)
        self.nbytes = {REAL: self.lib.fb_real_bytes(self._plan), HALF: self.lib.fb_half_bytes(self._plan),
                       FULL: self.lib.fb_full_bytes(self._plan)}
        self._amp_dense = None
        self._bins_key = None
        # ring of result records (see _ResultSlot): RES_SLOTS x RES_STRIDE doubles
        self._res_dev = None
        self._res_host = None
        self._res_next = 0          # serial of the next record to hand out
        self._res_fetched = 0       # records with serial < this are valid in _res_host (unless overwritten)
        self._res_waiters = {}      # serial -> weakref of the PendingSpectrum that wants the record
        _ENGINES.add(self)

    def close(self):
        if getattr(self, "_plan", None) is not None and self._plan:
            self.sync()
            self._work_half = None
            self._amp_dense = None
            self.release_idle_buffers()
            self._pool["closed"] = self.lib        # buffers of fields that outlive the engine are freed when dropped
            if self._res_dev:
                self.lib.fb_free(ctypes.c_void_p(self._res_dev))
                self._res_dev = None
            self.lib.fb_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory ---------------------------------------------------------------
    def _alloc_bytes(self, nbytes):
        free = self._pool.get(nbytes)
        if free:
            return _Buffer(free.pop(), nbytes, self._pool)
        p = ctypes.c_void_p()
        with _lib.on_device(self.device):                  # fb_malloc serves the current device
            try:
                _lib.call("fb_malloc", ctypes.byref(p), nbytes)
            except _lib.FastBoxError as e:
                if e.code != -4:                           # FB_ERR_NOMEM
                    raise
                # out of device memory: drop what nobody uses any more -- engines of boxes that are gone (reference
                # cycles keep them until a collection) and every live engine's cache of idle buffers -- and try once more
                import gc
                gc.collect()
                self.sync()
                for eng in list(_ENGINES):
                    eng.release_idle_buffers()
                _lib.call("fb_malloc", ctypes.byref(p), nbytes)
        return _Buffer(p.value, nbytes, self._pool)

    def release_idle_buffers(self):
        """Give the pooled (currently unused) device buffers back to the driver."""
        for key, ptrs in self._pool.items():
            while key != "closed" and ptrs:
                self.lib.fb_free(ctypes.c_void_p(ptrs.pop()))

    def empty(self, kind, as_complex=False):
        return DeviceArray(self, kind, self._alloc_bytes(self.nbytes[kind]), as_complex)

    def sync(self):
        _lib.call("fb_stream_sync", self.stream)

    def set_plane_batching(self, planes=-1, streams=0):
        """x-planes per batch of the y/z passes (-1: sized to the Infinity Cache for one box on the GPU; give each of
        several concurrently running boxes its share) and 1 | 2 streams for alternate batches (0: by grid size)."""
        _lib.call("fb_set_plane_batching", self._plan, int(planes), int(streams))

    def set_pass_schedule(self, plain=-1, generator=-1, binning=-1):
        """0: one workgroup per tile; 1: resident workgroups walking the tiles; -1: the library's choice by grid size --
        per class of strided FFT pass."""
        _lib.call("fb_set_pass_schedule", self._plan, int(plain), int(generator), int(binning))

    def set_tile_rows(self, nbytes=0):
        """Row segment of the strided passes' tiles at N = 2048 (single precision): 128 (default, 0) or 64 bytes."""
        _lib.call("fb_set_tile_rows", self._plan, int(nbytes))

    def upload(self, arr, kind):
        """Host ndarray (N,N,N) -> device.  kind REAL or FULL."""
        N = self.N
        a = np.asarray(arr)
        if a.shape != (N, N, N):
            raise ValueError("expected an array of shape %s, got %s" % ((N, N, N), a.shape))
        a = np.ascontiguousarray(a, dtype=self.rdtype if kind == REAL else self.cdtype)
        out = self.empty(kind)
        _lib.call("fb_memcpy_h2d", out.ptr, _ptr(a), a.nbytes, self.stream)
        return out

    def upload_raw(self, a):
        """Contiguous host array of any shape -> anonymous device buffer."""
        a = np.ascontiguousarray(a)
        buf = self._alloc_bytes(a.nbytes)
        _lib.call("fb_memcpy_h2d", buf.ptr, _ptr(a), a.nbytes, self.stream)
        return buf

    def download(self, d):
        N = self.N
        if d.kind == HALF:
            full = self.expand_half(d)
            return self.download(full)
        dt = self.rdtype if d.kind == REAL else self.cdtype
        h = np.empty((N, N, N), dtype=dt)
        _lib.call("fb_memcpy_d2h", _ptr(h), d.ptr, h.nbytes, self.stream)
        return h.astype(np.float64 if d.kind == REAL else np.complex128)

    def download_plane(self, d, ix):
        """One x-plane of a real device field as a host array in the plan's precision (no copy of the whole box)."""
        N = self.N
        if d.kind != REAL or not 0 <= ix < N:
            raise ValueError("download_plane: a real field and 0 <= ix < N")
        h = np.empty((N, N), dtype=self.rdtype)
        _lib.call("fb_memcpy_d2h", _ptr(h), d.ptr + ix * h.nbytes, h.nbytes, self.stream)
        _lib.call("fb_stream_sync", self.stream)
        return h

    def download_half_raw(self, d):
        """Half spectrum as stored, (N, N, pitch) complex; columns >= N/2+1 are padding."""
        h = np.empty((self.N, self.rows, self.pitch), dtype=self.cdtype)
        _lib.call("fb_memcpy_d2h", _ptr(h), d.ptr, h.nbytes, self.stream)
        return h[:, :self.N, :]

    def clone(self, d):
        out = self.empty(d.kind, d._as_complex)
        _lib.call("fb_memcpy_d2d", out.ptr, d.ptr, self.nbytes[d.kind], self.stream)
        return out

    # -- FFTs -------------------------------------------------------------------
    def fft_r2c(self, real, pre_exp=False):
        if pre_exp:
            self._set_exp_shift(0.0)
        out = self.empty(HALF)
        _lib.call("fb_fft_r2c", self._plan, real.ptr, out.ptr, 1 if pre_exp else 0, self.stream)
        return out

    def fft_c2r(self, half, scale=None, destroy=False, as_complex=False):
        """Re ifftn; numpy's 1/N^3 unless `scale` is given.  `half` is preserved unless destroy."""
        work = half if destroy else self.clone(half)
        out = self.empty(REAL, as_complex)
        _lib.call("fb_fft_c2r", self._plan, work.ptr, out.ptr,
                  float(scale if scale is not None else 1.0 / self.N ** 3), self.stream)
        return out

    def fft_c2c(self, full, direction, scale=1.0, inplace=False):
        out = full if inplace else self.clone(full)
        _lib.call("fb_fft_c2c", self._plan, out.ptr, int(direction), float(scale), self.stream)
        out.invalidate()
        return out

    def expand_half(self, half):
        out = self.empty(FULL)
        _lib.call("fb_expand_half", self._plan, half.ptr, out.ptr, self.stream)
        return out

    def crop_full(self, full):
        out = self.empty(HALF)
        _lib.call("fb_crop_full", self._plan, full.ptr, out.ptr, self.stream)
        return out

    # -- Gaussian realisation ----------------------------------------------------
    def set_amplitude_shells(self, amp):
        amp = np.ascontiguousarray(amp, dtype=np.float64)
        self._amp_dense = None
        _lib.call("fb_set_amplitude_shells", self._plan, amp.ctypes.data_as(_lib.P_double), amp.size)

    def set_amplitude_sym(self, amp):
        """sqrt(P boxfactor) per (|m_x|, |m_y|, |m_z|), shape (N/2+1,)*3: any box shape."""
        amp = np.ascontiguousarray(amp, dtype=np.float64)
        self._amp_dense = None
        _lib.call("fb_set_amplitude_sym", self._plan, amp.ctypes.data_as(_lib.P_double), amp.size)

    def set_amplitude_dense(self, amp_half):
        """amp_half: host (N, N, N/2+1) array of sqrt(P boxfactor)."""
        N = self.N
        padded = np.zeros((N, self.rows, self.pitch), dtype=self.rdtype)
        padded[:, :N, :N // 2 + 1] = amp_half
        self._amp_dense = self.upload_raw(padded)      # keep alive: the plan only borrows it
        _lib.call("fb_set_amplitude_dense", self._plan, self._amp_dense.ptr)

    def colour_noise(self, re, im):
        out = self.empty(HALF)
        _lib.call("fb_colour_noise", self._plan, re.ptr, im.ptr, out.ptr, self.stream)
        return out

    def colour_device(self, seed, realisation):
        out = self.empty(HALF)
        _lib.call("fb_colour_device", self._plan, int(seed) & (2 ** 64 - 1), int(realisation) & (2 ** 64 - 1),
                  out.ptr, self.stream)
        return out

    # -- P(k) ---------------------------------------------------------------------
    def set_bins(self, edges, thr=None, amb=()):
        edges = np.ascontiguousarray(edges, dtype=np.float64)
        key = (edges.tobytes(), None if thr is None else np.asarray(thr).tobytes(), tuple(amb))
        if key == self._bins_key:
            return
        if thr is not None:
            thr = np.ascontiguousarray(thr, dtype=np.int32)
            ambv = np.ascontiguousarray(list(amb) + [0], dtype=np.int32)
            _lib.call("fb_set_bins", self._plan, edges.ctypes.data_as(_lib.P_double), edges.size,
                      thr.ctypes.data_as(_lib.P_i32), ambv.ctypes.data_as(_lib.P_i32), len(amb))
        else:
            _lib.call("fb_set_bins", self._plan, edges.ctypes.data_as(_lib.P_double), edges.size, None, None, 0)
        self._bins_key = key
        self._nbins = edges.size
        self._cnt_cache = None

    def bin_power(self, spec, filt=None):
        """(count, sum |dk|^2, sum |dk|^4) per bin over the full grid; with ``filt`` = (kind, params) the
        sums are those of spec * T(k_perp, k_par), the filtered spectrum itself is never stored."""
        nb = self._nbins
        cnt, s1, s2 = (np.zeros(nb) for _ in range(3))
        outs = (cnt.ctypes.data_as(_lib.P_double), s1.ctypes.data_as(_lib.P_double), s2.ctypes.data_as(_lib.P_double))
        if filt is None:
            _lib.call("fb_bin_power", self._plan, spec.ptr, 1 if spec.kind == HALF else 0, *outs, self.stream)
        else:
            prm = (ctypes.c_double * 4)(*[float(x) for x in filt[1]])
            _lib.call("fb_bin_power_filtered", self._plan, spec.ptr, 1 if spec.kind == HALF else 0, int(filt[0]), prm,
                      None, *outs, self.stream)
        return cnt, s1, s2

    # -- k-space operators -----------------------------------------------------------
    def apply_filter(self, spec, kind, params=(0, 0, 0, 0), table=None, inplace=False):
        out = spec if inplace else self.empty(spec.kind)
        prm = (ctypes.c_double * 4)(*[float(x) for x in params])
        _lib.call("fb_apply_filter", self._plan, spec.ptr, out.ptr, 1 if spec.kind == HALF else 0, int(kind), prm,
                  table.ptr if table is not None else None, self.stream)
        out.invalidate()
        return out

    def axpby(self, x, y, a, b, c):
        """a x + b y + c over real cubes (y may be None), on the device."""
        x.ptr
        out = self.empty(REAL)
        _lib.call("fb_real_axpby", self._plan, x.ptr, y.ptr if y is not None else None, out.ptr, float(a), float(b),
                  float(c), self.stream)
        return out

    def multiply(self, x, y):
        out = self.empty(REAL)
        _lib.call("fb_real_multiply", self._plan, x.ptr, y.ptr, out.ptr, self.stream)
        return out

    def velocity_k(self, spec, comp, fac):
        out = self.empty(spec.kind)
        _lib.call("fb_velocity_k", self._plan, spec.ptr, out.ptr, 1 if spec.kind == HALF else 0, int(comp),
                  float(fac), self.stream)
        return out

    def potential_k(self, spec):
        out = self.empty(spec.kind)
        _lib.call("fb_potential_k", self._plan, spec.ptr, out.ptr, 1 if spec.kind == HALF else 0, self.stream)
        return out

    # -- real-space operators -----------------------------------------------------------
    def lognormal(self, real):
        out = self.empty(REAL)
        mean = ctypes.c_double()
        _lib.call("fb_lognormal", self._plan, real.ptr, out.ptr, ctypes.byref(mean), self.stream)
        return out, mean.value

    RSD_METHODS = {"linear": 0, "nearest": 1, "cubic": 2}       # FB_RSD_LINEAR, FB_RSD_NEAREST, FB_RSD_CUBIC

    def redshift_space(self, delta, vz, Hz, sigma_nl=0.0, noise=None, seed=0, method="linear"):
        out = self.empty(REAL)
        _lib.call("fb_redshift_space", self._plan, delta.ptr, vz.ptr, noise.ptr if noise is not None else None,
                  out.ptr, float(Hz), float(sigma_nl), int(seed) & (2 ** 64 - 1), self.RSD_METHODS[method], self.stream)
        return out

    # -- fused throughput path -------------------------------------------------------------
    def _scratch_half(self):
        if getattr(self, "_work_half", None) is None:
            self._work_half = self.empty(HALF)
        return self._work_half

    def realise_fused(self, seed, realisation):
        """Device-RNG Gaussian field; generator fused into the first inverse FFT pass."""
        out = self.empty(REAL)
        _lib.call("fb_realise_density_device", self._plan, int(seed) & (2 ** 64 - 1),
                  int(realisation) & (2 ** 64 - 1), self._scratch_half().ptr, out.ptr, self.stream)
        return out

    def realise_velocity_fused(self, seed, realisation, comp, fac):
        """Re ifftn(v_comp(k)) of the device-RNG realisation (seed, realisation), regenerated inside the
        first inverse FFT pass: no delta_k, no v(k) in memory."""
        out = self.empty(REAL)
        _lib.call("fb_realise_velocity_device", self._plan, int(seed) & (2 ** 64 - 1),
                  int(realisation) & (2 ** 64 - 1), int(comp), float(fac), self._scratch_half().ptr, out.ptr,
                  self.stream)
        return out

    def wait_for(self, other):
        """What is queued on this engine's stream from now on starts after everything queued on `other`'s stream so
        far has finished (no host wait).  Two boxes sharing a GPU use it to keep their generator passes apart."""
        with _lib.on_device(self.device):
            _lib.call("fb_stream_wait_stream", self.stream, other.stream)

    def realise_begin(self, seed, realisation):
        """Generator + x pass; returns the pending half spectrum (y and z passes still to do)."""
        pend = self.empty(HALF)
        _lib.call("fb_realise_density_begin", self._plan, int(seed) & (2 ** 64 - 1),
                  int(realisation) & (2 ** 64 - 1), pend.ptr, self.stream)
        return pend

    def realise_velocity_begin(self, seed, realisation, comp, fac):
        """Generator + x pass of velocity component `comp` of the realisation; returns the pending half spectrum
        (`realise_finish` runs its y and z passes, `power_redshift_space` consumes it)."""
        pend = self.empty(HALF)
        _lib.call("fb_realise_velocity_begin", self._plan, int(seed) & (2 ** 64 - 1), int(realisation) & (2 ** 64 - 1),
                  int(comp), float(fac), pend.ptr, self.stream)
        return pend

    @property
    def fuses_redshift_space(self):
        """Does `power_redshift_space` exist for this plan?  (single precision, 64 <= N <= 512)"""
        return self.precision == "f32" and 64 <= self.N <= 512 and (self.N & (self.N - 1)) == 0 \
            and not os.environ.get("FASTBOX_NO_RSD_FUSION")          # (tuning aid: time the separate kernels)

    def power_redshift_space(self, pend_delta, pend_vz, Hz, sigma_nl, seed, method, filt=None, field=False, keep_field=True):
        """P(k) of the redshift-space density of two pending realisations (density, v_z), optionally through a
        k_perp / k_par filter (kind, params): one z pass does both inverse transforms, the remap and the forward
        transform.  Returns (results buffer, delta_x or None, work half spectrum: the filtered spectrum -- with `field`
        transformed back along x, for `fft_c2r_yz` -- or scratch when there is no filter).  Both pendings are destroyed."""
        res = self._result_slot()
        real = self.empty(REAL) if keep_field else None
        out = self.empty(HALF)
        kind = -1 if filt is None else int(filt[0])
        prm = (ctypes.c_double * 4)(*[float(x) for x in (filt[1] if filt is not None else (0, 0, 0, 0))])
        _lib.call("fb_power_spectrum_redshift_space", self._plan, pend_delta.ptr, pend_vz.ptr,
                  real.ptr if keep_field else None, out.ptr, float(Hz), float(sigma_nl), int(seed) & (2 ** 64 - 1),
                  self.RSD_METHODS[method], kind, prm, None, 1 if field else 0, res.ptr, self.stream)
        return res, real, out

    def realise_finish(self, pend):
        out = self.empty(REAL)
        _lib.call("fb_realise_density_finish", self._plan, pend.ptr, out.ptr, self.stream)
        return out

    def _set_exp_shift(self, shift):
        if shift != getattr(self, "_exp_shift", 0.0):
            _lib.call("fb_set_exp_shift", self._plan, float(shift))
            self._exp_shift = float(shift)

    def power_pending(self, pend, pre_exp=False, exp_shift=0.0, keep_field=True):
        """Fused z pass (writes delta_x unless keep_field is False) + P(k) of (exp of) it.  Returns (results buffer,
        delta_x or None).  exp_shift: the exponentials are formed as exp(x - exp_shift) (fb_set_exp_shift)."""
        self._set_exp_shift(exp_shift if pre_exp else getattr(self, "_exp_shift", 0.0))
        res = self._result_slot()
        out = self.empty(REAL) if keep_field else None
        _lib.call("fb_power_spectrum_pending", self._plan, pend.ptr, out.ptr if keep_field else None,
                  1 if pre_exp else 0, res.ptr, self.stream)
        return res, out

    def montecarlo_power(self, seed, first, count, stride=1, pre_exp=False, exp_shift=0.0):
        """`count` realisations (seed, first + i stride) drawn and estimated by ONE library call (fb_montecarlo_power: the
        launches of realise_begin + power_pending(keep_field=False) per realisation, back to back).  Returns the host
        array [count][2 nbins + 1] of their bin sums (waits for the stream)."""
        self._set_exp_shift(exp_shift if pre_exp else getattr(self, "_exp_shift", 0.0))
        nb = self._nbins
        width = 2 * nb + 1
        out = np.empty((int(count), width))
        if count:
            res = self._alloc_bytes(int(count) * width * 8)
            work = self._scratch_half()
            _lib.call("fb_montecarlo_power", self._plan, int(seed) & (2 ** 64 - 1), int(first) & (2 ** 64 - 1), int(stride),
                      int(count), work.ptr, None, 1 if pre_exp else 0, res.ptr, width, self.stream)
            _lib.call("fb_memcpy_d2h", _ptr(out), res.ptr, out.nbytes, self.stream)
        return out

    def power_fused(self, real, pre_exp=False, keep_spectrum=False, exp_shift=0.0):
        """Asynchronous r2c + shell binning (cubic boxes).  Returns (results buffer, spectrum or None);
        results = [2*nbins+1] doubles on the device, fetched with `fetch_results`."""
        self._set_exp_shift(exp_shift if pre_exp else getattr(self, "_exp_shift", 0.0))
        res = self._result_slot()
        work = self.empty(HALF) if keep_spectrum else self._scratch_half()
        _lib.call("fb_power_spectrum_device", self._plan, real.ptr, work.ptr, 1 if pre_exp else 0,
                  1 if keep_spectrum else 0, res.ptr, self.stream)
        return res, (work if keep_spectrum else None)

    def power_filtered(self, real, filt, field=False):
        """Asynchronous r2c of `real` whose last pass multiplies by the filter (kind, params), keeps the
        filtered spectrum and bins it.  Returns (results buffer, filtered half spectrum).  With ``field`` the last
        pass also takes the inverse transform of every x line it has filtered: the buffer returned is then the
        filtered spectrum half way back to the field, for `fft_c2r_yz` only."""
        res = self._result_slot()
        out = self.empty(HALF)
        prm = (ctypes.c_double * 4)(*[float(x) for x in filt[1]])
        _lib.call("fb_power_spectrum_filtered_field" if field else "fb_power_spectrum_filtered", self._plan, real.ptr,
                  out.ptr, int(filt[0]), prm, None, res.ptr, self.stream)
        return res, out

    def fft_c2r_yz(self, half_x_done, as_complex=False):
        """The y and z passes of `fft_c2r` (numpy's 1/N^3) for `power_filtered(..., field=True)`'s buffer, which is
        destroyed."""
        out = self.empty(REAL, as_complex)
        _lib.call("fb_fft_c2r_yz", self._plan, half_x_done.ptr, out.ptr, 1.0 / self.N ** 3, self.stream)
        return out

    RES_SLOTS, RES_STRIDE = 256, 2 * 256 + 1      # 256 = FB_MAX_BINS

    def _result_slot(self):
        """Next record of the result ring.  A record is overwritten RES_SLOTS hand-outs later; whatever is
        still unfetched by then is brought to the host first."""
        if self._res_dev is None:
            p = ctypes.c_void_p()
            with _lib.on_device(self.device):
                _lib.call("fb_malloc", ctypes.byref(p), self.RES_SLOTS * self.RES_STRIDE * 8)
            self._res_dev = p.value
            self._res_host = np.zeros((self.RES_SLOTS, self.RES_STRIDE))
        if self._res_next - self._res_fetched >= self.RES_SLOTS:
            self._fetch_pending_results()
        serial = self._res_next
        self._res_next += 1
        return _ResultSlot(self._res_dev + (serial % self.RES_SLOTS) * self.RES_STRIDE * 8, serial)

    def _fetch_pending_results(self):
        """One copy (two if the ring wraps) for every record queued since the last fetch; waits for the stream."""
        lo, hi = self._res_fetched, self._res_next
        a, b = lo % self.RES_SLOTS, hi % self.RES_SLOTS
        spans = [(a, b)] if (a < b) else [(a, self.RES_SLOTS), (0, b)]
        for x, y in spans:
            if y > x:
                _lib.call("fb_memcpy_d2h", _ptr(self._res_host[x:y]), self._res_dev + x * self.RES_STRIDE * 8,
                          (y - x) * self.RES_STRIDE * 8, self.stream)
        self._res_fetched = hi
        batch = []
        for serial in range(lo, hi):             # hand every fetched record to whoever waits for it
            w = self._res_waiters.pop(serial, None)
            owner = w() if w is not None else None
            if owner is not None:
                owner._raw = self._res_host[serial % self.RES_SLOTS].copy()
                batch.append(owner)
        # a Monte-Carlo loop resolves hundreds of spectra at once: their host arithmetic (range check, mean / spread per
        # bin: a dozen numpy calls on 20-element arrays, ~25 us each time) is done for the whole batch in one go
        if len(batch) > 1:
            hook = getattr(type(batch[0]), "_finish_batch", None)
            if hook is not None:
                hook(batch)

    def register_waiter(self, res, owner):
        """`owner._raw` receives the record of `res` when it is fetched (possibly as part of a batch)."""
        if isinstance(res, _ResultSlot):
            self._res_waiters[res.serial] = weakref.ref(owner)

    def fetch_results(self, res, nbins, owner=None):
        if isinstance(res, _ResultSlot):
            if owner is not None and getattr(owner, "_raw", None) is None:
                self._fetch_pending_results()
            h = owner._raw if owner is not None and getattr(owner, "_raw", None) is not None else None
            if h is None:                          # no registered owner: read the ring's host mirror
                if res.serial >= self._res_fetched:
                    self._fetch_pending_results()
                if self._res_next - res.serial > self.RES_SLOTS:
                    raise RuntimeError("result record overwritten (not registered with register_waiter)")
                h = self._res_host[res.serial % self.RES_SLOTS]
        else:
            h = np.empty(2 * nbins + 1)
            _lib.call("fb_memcpy_d2h", _ptr(h), res.ptr, h.nbytes, self.stream)
        return h[0:2 * nbins:2].copy(), h[1:2 * nbins:2].copy(), float(h[2 * nbins])

    def bin_counts(self):
        """Modes per bin of the current bin set (read-only array, computed once per fb_set_bins)."""
        c = getattr(self, "_cnt_cache", None)
        if c is None:
            c = np.zeros(self._nbins)
            _lib.call("fb_bin_counts", self._plan, c.ctypes.data_as(_lib.P_double))
            c.setflags(write=False)
            self._cnt_cache = c
        return c

    # -- profiling ------------------------------------------------------------------------
    PROF_NAMES = ("fft_strided", "fft_contig", "colour", "bin", "filter", "velpot", "realop", "rsd", "layout",
                  "fft_gen", "fft_bin", "pca")

    def profile_start(self, only=None, stride=1):
        """Bracket kernel launches with HIP events; `only` = iterable of class names to restrict to; `stride` =
        bracket every stride-th of those launches only (each event pair costs stream time)."""
        mask = 0xFFFFFFFF if only is None else sum(1 << self.PROF_NAMES.index(n) for n in only)
        _lib.call("fb_profile_select", self._plan, mask)
        _lib.call("fb_profile_sample", self._plan, int(stride), None)
        _lib.call("fb_profile_start", self._plan)

    def profile_seen(self):
        """Selected launches since profile_start (bracketed or not)."""
        seen = ctypes.c_int64()
        _lib.call("fb_profile_sample", self._plan, 0, ctypes.byref(seen))       # 0: the sampling stride stays as it is
        return seen.value

    def profile_stop(self):
        """{kernel class: (total ms, launches)} measured with HIP events on the launch stream."""
        n = len(self.PROF_NAMES)
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_int64 * n)()
        _lib.call("fb_profile_stop", self._plan, self.stream, ms, cnt, n)
        return {name: (ms[i], cnt[i]) for i, name in enumerate(self.PROF_NAMES)}

    def sum_real(self, real, squared=False):
        v = ctypes.c_double()
        _lib.call("fb_sum_real", self._plan, real.ptr, 1 if squared else 0, ctypes.byref(v), self.stream)
        return v.value

    def max_real(self, real):
        """Largest finite value of a real field (waits for the stream)."""
        v = ctypes.c_double()
        _lib.call("fb_max_real", self._plan, real.ptr, ctypes.byref(v), self.stream)
        return v.value

    def sumsq_half(self, half):
        v = ctypes.c_double()
        _lib.call("fb_sumsq_half", self._plan, half.ptr, ctypes.byref(v), self.stream)
        return v.value
