"""
``CosmoBox`` -- drop-in for the density-field hot path of ``fastbox.box.CosmoBox``
(philbull/FastBox, fastbox/box.py) with the arithmetic running as HIP kernels
on an MI355X behind libfastbox_hip.so.

Same constructor, method names, argument order, defaults and exceptions as the
reference; fields are returned as ``DeviceArray`` objects that stay in HBM and
turn into float64 / complex128 ndarrays when numpy touches them.  Additive,
reference-preserving keyword arguments: ``precision`` ('f32' storage with fp64
bin sums, or 'f64'), ``rng`` ('numpy' = the reference's legacy global stream,
drawn on the host and uploaded -- same seed, same field; 'device' = on-device
counter-based Philox4x32-10 RNG for throughput, reproducible on the host with
``fastbox_amd.rng``), ``seed``, ``device``, ``stream``.

No CPU fallback exists: without the HIP library or a GPU every compute method
raises.
"""
import weakref

import numpy as np

from . import hostgeom

from . import cosmology as _builtin_cosmology
from . import device as _dev
from ._lib import FB_FILT_TABLE, FB_FILT_TOPHAT
from .device import DeviceArray, Engine, FULL, HALF, REAL
from .transfer import DeviceFilter

try:                                        # the reference's provider, if present
    import pyccl as _ccl
except Exception:                           # not installed in this image
    _ccl = _builtin_cosmology

# Speed of light (m/s)
C = 299792458.

default_cosmo = dict(Omega_c=0.25, Omega_b=0.05, h=0.7, n_s=0.95, sigma8=0.8,
                     transfer_function='eisenstein_hu')

try:
    from scipy.integrate import simpson as _simpson
except Exception:                           # pragma: no cover
    from scipy.integrate import simps as _simpson


def _finish_bins(cnt, s1, s2, boxfactor, eps=0.):
    """(mean, std/sqrt(n)) per bin from (count, sum |dk|^2, sum |dk|^4), bin 0 dropped
    (box.py:761-768); see hostgeom.finish_bins."""
    return hostgeom.finish_bins(cnt, s1, s2, boxfactor, eps)


def _eps_of(engine):
    """rounding unit of the plan's |delta_k|^2 values: the precision floor of the variance form"""
    return 2. ** -23 if engine.precision == "f32" else 2. ** -52


class _Ready(object):
    def __init__(self, out):
        self._out = out

    def result(self):
        return self._out


class PendingSpectrum(object):
    """Handle on a power spectrum whose bin sums are still being computed on the device."""

    def __init__(self, engine, res, nbins, kc, boxfactor, ln_voxels, keepalive, redo=None):
        self._eng, self._res, self._nb, self._kc, self._bf = engine, res, nbins, kc, boxfactor
        self._lnv, self._keep, self._out = ln_voxels, keepalive, None
        self._cnt = engine.bin_counts()
        self._raw = None                       # the device record, delivered by the engine's batched fetch
        self._redo = redo                      # log-normal: () -> (s1, s2, esum) with the exact shift (see result)
        self.repeated = False
        engine.register_waiter(res, self)

    def _in_range(self, s1, s2, esum):
        return hostgeom.lognormal_sums_in_range(self._cnt, s1, s2, esum)

    @staticmethod
    def _finish_batch(owners):
        """The host arithmetic of `result()` for every record of one fetch at once (same numpy expressions, broadcast over
        the records: identical numbers).  Records whose log-normal sums are out of range are left to `result()`, which
        repeats the step."""
        groups = {}
        for o in owners:
            if type(o) is PendingSpectrum and o._out is None and o._raw is not None:
                groups.setdefault((o._nb, id(o._cnt), o._lnv, o._bf, _eps_of(o._eng)), []).append(o)
        for (nb, _, lnv, bf, eps), grp in groups.items():
            if len(grp) < 2:
                continue
            R = np.stack([o._raw for o in grp])
            s1, s2, esum = R[:, 0:2 * nb:2], R[:, 1:2 * nb:2], R[:, 2 * nb]
            cnt = grp[0]._cnt
            good = np.ones(len(grp), dtype=bool)
            with np.errstate(all="ignore"):
                if lnv:
                    good = hostgeom.lognormal_sums_in_range_many(cnt, s1, s2, esum)
                    mean = esum / lnv
                    s1, s2 = hostgeom.lognormal_rescale(s1, s2, mean)
                vals, std = hostgeom.finish_bins_many(cnt, s1, s2, bf, eps)
            for i, o in enumerate(grp):
                if good[i]:
                    o._out = (o._kc, vals[i].copy(), std[i].copy())
                    o._res = o._keep = o._redo = None

    def result(self):
        if self._out is None:
            s1, s2, esum = self._eng.fetch_results(self._res, self._nb, owner=self)
            if self._lnv:                      # transform of exp(d - shift): rescale to exp(d)/mean - 1
                if not self._in_range(s1, s2, esum):
                    # the shift was taken from the field's variance (hostgeom.lognormal_shift) and this realisation's
                    # largest values lie outside what that allows for: once more, with the shift from its maximum
                    if self._redo is None:
                        raise FloatingPointError("log-normal P(k): the exponentials left the plan's floating-point "
                                                 "range (sum = %r)" % (esum,))
                    s1, s2, esum = self._redo()
                    self.repeated = True
                    if not self._in_range(s1, s2, esum):
                        raise FloatingPointError("log-normal P(k): non-finite bin sums (sum of exponentials %r); "
                                                 "is the field finite?" % (esum,))
                mean = esum / self._lnv
                s1, s2 = hostgeom.lognormal_rescale(s1, s2, mean)
            self._out = (self._kc,) + _finish_bins(self._cnt, s1, s2, self._bf, _eps_of(self._eng))
            self._res = self._keep = self._redo = None
        return self._out


class LognormalField(DeviceArray):
    """exp(d)/<exp(d)> - 1 of a device field, materialised on first read."""

    def __init__(self, engine, source):
        DeviceArray.__init__(self, engine, REAL, None)
        self.source = source

    @property
    def materialised(self):
        return self._buf is not None

    @property
    def ptr(self):
        if self._buf is None:
            out, _ = self.engine.lognormal(self.source)
            self._buf = out._buf
        return self._buf.ptr


class LazySpectrum(DeviceArray):
    """A k-space field computed by `make()` on first use (e.g. one velocity component: callers of the
    reference usually transform only v_z, box.py:285 returns all three)."""

    def __init__(self, engine, kind, make, recipe=None, source_real=None):
        DeviceArray.__init__(self, engine, kind, None)
        self._make = make
        self.recipe = recipe      # (amp_key, seed, realisation, comp, fac) of a device-RNG velocity component
        self.source_real = source_real    # to_k(field): the real field this is the (pending) transform of

    @property
    def materialised(self):
        return self._buf is not None

    @property
    def ptr(self):
        if self._buf is None:
            self._buf = self._make()._buf
            self._make = None
        return self._buf.ptr


class FilteredField(DeviceArray):
    """ifftn(field_k * T(k_perp, k_par)) for a Hermitian device spectrum and a k_par-even ``DeviceFilter``
    (apply_transfer_fn, box.py:356-381), formed when it is first read.  ``binned_power_spectrum`` of it
    (or of its ``.real``) needs no transform of its own: it bins |field_k T|^2 from ``field_k``; when
    ``field_k`` is itself a pending ``CosmoBox.to_k(field)``, the multiply and the binning happen inside the last (x)
    pass of that forward transform, which also takes the inverse transform of each x line it has filtered -- reading
    the field afterwards costs the y and z passes only."""

    def __init__(self, engine, spectrum, filt, as_complex=True, parent=None):
        DeviceArray.__init__(self, engine, REAL, None, as_complex)
        self.spectrum, self.filter, self._parent = spectrum, filt, parent
        self._x_done = None              # field_k * T with its x lines already transformed back, once a fused P(k) has
                                         # produced it (see ptr)

    def _root(self):
        return self if self._parent is None else self._parent._root()

    @property
    def materialised(self):
        return self._buf is not None

    @property
    def ptr(self):
        if self._buf is None:
            root = self._root()
            if root is not self:
                root.ptr
                self._buf = root._buf
            elif self._x_done is not None:
                # the fused P(k) pass took the inverse x transform of every line it filtered: y and z remain
                self._buf = self.engine.fft_c2r_yz(self._x_done)._buf
                self._x_done = None
            else:
                dk = self.engine.apply_filter(self.spectrum, self.filter.kind, self.filter.params)
                self._buf = self.engine.fft_c2r(dk, destroy=True)._buf
        return self._buf.ptr

    @property
    def real(self):
        return FilteredField(self.engine, self.spectrum, self.filter, as_complex=False, parent=self)


class PendingDensity(DeviceArray):
    """delta_x of a device-RNG realisation whose last (z) FFT pass has not run yet.  Reading it
    runs that pass; ``binned_power_spectrum(delta_x=...)`` of it (or of its log-normal) instead
    fuses the pass with the power spectrum's first one, and fills in delta_x on the way."""

    def __init__(self, engine, pending_half, generator=None, regenerate=None, sigma2=None):
        DeviceArray.__init__(self, engine, REAL, None)
        self.sigma2 = sigma2            # variance of the distribution the field was drawn from (log-normal shift)
        self._pending = pending_half
        self.generator = generator      # (amp_key, seed, realisation): enough to regenerate delta_k
        self._regenerate = regenerate   # () -> real DeviceArray of the same realisation (counter-based generator)

    @property
    def materialised(self):
        return self._buf is not None

    def _adopt(self, real):
        self._buf, self._pending = real._buf, None

    @property
    def ptr(self):
        if self._buf is None:
            if self._pending is not None:
                self._adopt(self.engine.realise_finish(self._pending))
            else:
                # the pending spectrum went into a power spectrum that was asked not to keep the field
                # (binned_power_spectrum(..., keep_field=False)): draw the same realisation again
                self._adopt(self._regenerate())
        return self._buf.ptr


class RedshiftSpaceField(DeviceArray):
    """``redshift_space_density`` of two device-generator realisations (density, v_z) whose last FFT passes have not run
    yet.  Reading it finishes both and runs the remap kernel; asking for the power spectrum of its transform -- directly,
    or through ``apply_transfer_fn(to_k(...))`` -- instead runs ONE z pass per line of sight that finishes both inverse
    transforms, remaps and starts the forward transform (``Engine.power_redshift_space``): v_z in real space and this
    field itself are then never written.  The numbers are the same bit for bit."""

    def __init__(self, engine, delta, vz, args):
        DeviceArray.__init__(self, engine, REAL, None)
        self.delta, self.vz, self.args = delta, vz, args      # args = (Hz, sigma_nl, seed, method)

    @property
    def materialised(self):
        return self._buf is not None

    def fusable(self):
        return self._buf is None and self.delta is not self.vz and all(
            isinstance(f, PendingDensity) and not f.materialised and f._pending is not None for f in (self.delta, self.vz))

    def consume(self, filt=None, field=False):
        """The fused pass: (results buffer, work half spectrum); delta_x is filled in, v_z's pending spectrum is gone
        (reading v_z later draws it again)."""
        Hz, sigma_nl, seed, method = self.args
        res, real, work = self.engine.power_redshift_space(self.delta._pending, self.vz._pending, Hz, sigma_nl, seed, method,
                                                           filt=filt, field=field)
        self.delta._adopt(real)
        self.vz._pending = None
        return res, work

    @property
    def ptr(self):
        if self._buf is None:
            Hz, sigma_nl, seed, method = self.args
            self._buf = self.engine.redshift_space(self.delta, self.vz, Hz, sigma_nl, None, seed, method)._buf
        return self._buf.ptr


class CosmoBox(object):

    def __init__(self, cosmo, box_scale=1e3, nsamp=32, redshift=0.,
                 line_freq=1420.405752, realise_now=True,
                 precision="f32", rng="numpy", seed=0, device=0, stream=None):
        # box.py:61-64
        if isinstance(cosmo, dict):
            cosmo = _ccl.Cosmology(**cosmo)
        if not isinstance(cosmo, _ccl.Cosmology):
            raise TypeError("`cosmo` must be a CCL Cosmology object or dict.")
        self.cosmo = cosmo
        self.N = nsamp
        self.redshift = redshift
        self.scale_factor = 1. / (1. + redshift)
        self.line_freq = line_freq

        # grid including both end points, box.py:76-89
        if isinstance(box_scale, tuple):
            assert len(box_scale) == 3, "Must specify scale of x, y, z dimensions"
            self.x, self.y, self.z = [np.linspace(-0.5 * s, 0.5 * s, nsamp) for s in box_scale]
        else:
            self.x = self.y = self.z = np.linspace(-0.5 * box_scale, 0.5 * box_scale, nsamp)
        self.Lx = self.x[-1] - self.x[0]
        self.Ly = self.y[-1] - self.y[0]
        self.Lz = self.z[-1] - self.z[0]
        self.boxfactor = (self.N ** 6.) / (self.Lx * self.Ly * self.Lz)      # box.py:94
        self.kmin = 2. * np.pi / np.max([self.Lx, self.Ly, self.Lz])          # box.py:100
        self.kmax = 2. * np.pi * np.sqrt(3.) * self.N / np.min([self.Lx, self.Ly, self.Lz])

        if rng in ("threefry", "philox"):          # historical spellings of the device generator
            rng = "device"
        if rng not in ("numpy", "device"):
            raise ValueError("rng must be 'numpy' or 'device'")
        self.rng, self.seed, self._realisation = rng, int(seed), 0
        # grids that are not powers of two (round 4: even, prime factors 2, 3, 5, up to 1024 -- numpy's FFT takes any nsamp,
        # box.py:25-26): the library's plain FFT passes; nothing is fused into them, so such a box takes the step-by-step
        # routes a non-cubic box takes (exact |k| per mode instead of shell tables, stored spectra)
        self._plain = (self.N & (self.N - 1)) != 0
        self._cubic = (self.Lx == self.Ly == self.Lz) and not self._plain
        self._grids = None
        self._amp_key = None
        self._bin_cache = {}
        self._delta_k = None
        self.lognormal_repeats = 0        # fused log-normal spectra formed a second time with the exact shift
        self.set_fft_sample_spacing()
        self.engine = Engine(self.N, (self.Lx, self.Ly, self.Lz), self._axis2, self._ksc, self._kpar, self.z,
                             precision=precision, device=device, stream=stream)

        if realise_now:                                                       # box.py:104-107
            self.realise_density()
            self.realise_velocity()
            self.realise_potential()

    # ------------------------------------------------------------------ k grids
    def set_fft_sample_spacing(self):
        """1-D tables behind |k|, k_perp, k_par (box.py:119-127, 254-256, 374-375).  The
        (N,N,N) arrays Kx, Ky, Kz, k of the reference are never built on the device; the
        attributes of the same name materialise them on the host on first access."""
        N = self.N
        m = hostgeom.mode_numbers(N)        # the reference's own numbering, index-as-value quirk at non-powers of two included
        self._modes = m
        L = (self.Lx, self.Ly, self.Lz)
        self._axis2 = np.concatenate([(m / l) ** 2. for l in L])
        self._ksc = np.concatenate([m * (2. * np.pi / l) for l in L])
        self._kpar = 2. * np.pi * m / self.Lz
        self._grids = None

    def _host_grids(self):
        if self._grids is None:
            N, m = self.N, self._modes
            ones = np.ones((N, N, N))
            Kx, Ky, Kz = m[:, None, None] * ones, m[None, :, None] * ones, m[None, None, :] * ones
            k = 2. * np.pi * np.sqrt((Kx / self.Lx) ** 2. + (Ky / self.Ly) ** 2. + (Kz / self.Lz) ** 2.)
            self._grids = (Kx, Ky, Kz, k)
        return self._grids

    Kx = property(lambda self: self._host_grids()[0])
    Ky = property(lambda self: self._host_grids()[1])
    Kz = property(lambda self: self._host_grids()[2])
    k = property(lambda self: self._host_grids()[3])

    # --------------------------------------------------------------- conversions
    def _as_real(self, a):
        if isinstance(a, DeviceArray):
            if a.kind != REAL:
                raise TypeError("expected a real-space field")
            return a
        a = np.asarray(a)
        if np.iscomplexobj(a):
            a = a.real
        return self.engine.upload(a, REAL)

    def _as_spectrum(self, a):
        """DeviceArray half/full, or a host complex (N,N,N) array uploaded as 'full'."""
        if isinstance(a, DeviceArray):
            if a.kind == REAL:
                raise TypeError("expected a Fourier-space field")
            return a
        return self.engine.upload(np.asarray(a), FULL)

    @property
    def delta_k(self):
        """fftn(delta_x) (box.py:193), computed on first use after a realisation."""
        if self._delta_k is None:
            if getattr(self, "delta_x", None) is None:
                raise AttributeError("'CosmoBox' object has no attribute 'delta_k'")
            self._delta_k = self.engine.fft_r2c(self.delta_x)
        return self._delta_k

    @delta_k.setter
    def delta_k(self, value):
        self._delta_k = None if value is None else self._as_spectrum(value)

    def to_k(self, field):
        """fftn(field) of a real field as a device half spectrum (what callers of the reference write as
        ``np.fft.fftn(field)``), formed on first use: ``apply_transfer_fn`` + ``binned_power_spectrum`` of
        it run as one forward transform with the filter and the binning inside its last pass."""
        real = self._as_real(field)
        return LazySpectrum(self.engine, HALF, lambda: self.engine.fft_r2c(real), source_real=real)

    def to_real(self, field_k):
        """Real part of ifftn(field_k) as a device field (what callers of the reference
        write as ``np.fft.ifftn(box.velocity_k[2]).real``)."""
        if isinstance(field_k, LazySpectrum) and field_k.recipe is not None and not field_k.materialised:
            # velocity of a device-RNG realisation: regenerate delta_k inside the first inverse pass
            amp_key, seed, realisation, comp, fac = field_k.recipe
            self._set_amplitude(*amp_key)
            if comp == 2 and self.engine.fuses_redshift_space:
                # v_z: the last passes are deferred -- redshift_space_density of it can then finish them inside its own
                # kernel (RedshiftSpaceField); reading the field runs them as before
                wbox = weakref.ref(self)

                def again():
                    box = wbox()
                    if box is None:
                        raise RuntimeError("the CosmoBox that drew this field is gone; it cannot be drawn again")
                    box._set_amplitude(*amp_key)
                    return box.engine.realise_velocity_fused(seed, realisation, comp, fac)
                return PendingDensity(self.engine, self.engine.realise_velocity_begin(seed, realisation, comp, fac),
                                      regenerate=again)
            return self.engine.realise_velocity_fused(seed, realisation, comp, fac)
        f = self._as_spectrum(field_k)
        if f.kind == HALF:
            return self.engine.fft_c2r(f)
        full = self.engine.fft_c2c(f, +1, 1.0 / self.N ** 3)
        return self.engine.upload(full.host().real, REAL)

    # --------------------------------------------------------- Gaussian realisation
    def _power(self, k, scale_factor, linear):
        fn = _ccl.linear_matter_power if linear else _ccl.nonlin_matter_power
        return fn(self.cosmo, k=k, a=scale_factor)

    def _set_amplitude(self, scale_factor, linear):
        """sqrt(nan_to_num(P(k)) * boxfactor), box.py:161-171, on unique |k| shells for a
        cubic box, per stored mode otherwise."""
        key = (float(scale_factor), bool(linear))
        if key == self._amp_key:
            return
        N = self.N
        with np.errstate(all="ignore"):
            if self._cubic:
                n2 = np.arange(3 * (N // 2) ** 2 + 1, dtype=np.float64)
                k = 2. * np.pi * np.sqrt(n2) / self.Lx
                pk = np.nan_to_num(np.asarray(self._power(k, scale_factor, linear), dtype=np.float64))
                amp = np.sqrt(pk * self.boxfactor)
                self.engine.set_amplitude_shells(amp)
                self._sigma2 = hostgeom.field_variance_cubic(N, amp)       # variance of the fields drawn from it
            else:
                # |k| depends on the mode numbers only through |m_x|, |m_y|, |m_z|: evaluate P(k) on the
                # (N/2+1)^3 magnitudes (an eighth of the stored modes), in the reference's operation order
                M = N // 2 + 1
                a = self._axis2
                s = (a[:M, None, None] + a[N:N + M][None, :, None]) + a[2 * N:2 * N + M][None, None, :]
                k = 2. * np.pi * np.sqrt(s)
                pk = np.nan_to_num(np.asarray(self._power(k.flatten(), scale_factor, linear), dtype=np.float64))
                amp = np.sqrt(pk.reshape(k.shape) * self.boxfactor)
                self.engine.set_amplitude_sym(amp)
                self._sigma2 = hostgeom.field_variance_sym(N, amp)
        self._amp_key = key

    def realise_density(self, linear=False, redshift=None, inplace=True):
        """Gaussian random field with the matter power spectrum (box.py:130-194).

        rng='numpy': ``re`` then ``im`` are drawn from the legacy global numpy stream
        exactly as the reference does, so ``np.random.seed(s)`` gives the same field.
        Conscious fix: with ``inplace=False`` the reference transforms a *stale*
        ``self.delta_k`` (box.py:187); here the new field is returned.
        """
        if redshift is None:
            redshift = self.redshift
        scale_factor = 1. / (1. + redshift)
        self._set_amplitude(scale_factor, linear)
        eng = self.engine
        N = self.N
        if self.rng == "numpy":
            re = eng.upload(np.random.normal(0.0, 1.0, (N, N, N)), REAL)
            im = eng.upload(np.random.normal(0.0, 1.0, (N, N, N)), REAL)
            half = eng.colour_noise(re, im)
            del re, im
            delta_x = eng.fft_c2r(half, destroy=True)
            delta_x.sigma2 = self._sigma2
        elif self._plain:
            # no fused generator pass on such a grid: the coloured half spectrum (same counters, same numbers), then c2r
            delta_x = eng.fft_c2r(eng.colour_device(self.seed, self._realisation), destroy=True)
            delta_x.sigma2 = self._sigma2
            self.last_realisation = self._realisation
            self._realisation += 1
        else:
            # generator fused into the first inverse FFT pass (no coloured spectrum round trip); the
            # last pass is deferred so that a following P(k) can fuse it with its own first pass
            amp_key, seed, real = self._amp_key, self.seed, self._realisation
            wbox = weakref.ref(self)      # (no reference cycle box -> delta_x -> closure -> box: a dropped field must
                                          # give its buffer back at once, not at the next garbage collection)

            def again():
                box = wbox()
                if box is None:
                    raise RuntimeError("the CosmoBox that drew this field is gone; it cannot be drawn again")
                box._set_amplitude(*amp_key)
                return box.engine.realise_fused(seed, real)
            delta_x = PendingDensity(eng, eng.realise_begin(seed, real), generator=(amp_key, seed, real),
                                     regenerate=again, sigma2=self._sigma2)
            self.last_realisation = self._realisation
            self._realisation += 1
        if inplace:
            if redshift != self.redshift:
                print("Warning: Storing density field into self.delta_x with a "
                      "different redshift than self.redshift.")
            self.delta_x = delta_x
            self._delta_k = None          # = fftn(delta_x), materialised on demand
        return delta_x

    def realisation_spectra(self, count, nbins=20, kbins=None, lognormal=False, linear=False, redshift=None, stride=1):
        """Additive (a Monte-Carlo loop in one call): P(k) of the next `count` realisations of this box's device generator
        -- indices r, r + stride, ... from the box's counter -- as ``(kc, pk[count, nbins-1], stddev[count, nbins-1])``.
        The same numbers, realisation by realisation, as

            dx = box.realise_density(); box.binned_power_spectrum(delta_x=box.lognormal(dx) if lognormal else dx, nbins=nbins)

        (box.py:130-194, :441-460, :696-768), but queued by the library itself (fb_montecarlo_power), which is what a
        small box's step is bound by; the fields are not kept (``box.delta_x`` is left as it was).  Cubic boxes on a
        power-of-two grid with rng='device'; anything else runs the loop above."""
        count, stride = int(count), int(stride)
        if count < 0 or stride < 1:
            raise ValueError("count >= 0 and stride >= 1")
        if redshift is None:
            redshift = self.redshift
        bins, kc, thr, amb = self._bin_setup(nbins, kbins)
        first = self._realisation
        fast = self.rng == "device" and not self._plain and thr is not None and (not lognormal or bins[0] > 0.)

        def one_by_one(indices):
            rows = []
            for r in indices:
                self._realisation = r
                dx = self.realise_density(linear=linear, redshift=redshift, inplace=False)
                rows.append(self.binned_power_spectrum(delta_x=self.lognormal(dx) if lognormal else dx, nbins=nbins, kbins=kbins,
                                                       keep_field=False))
            return rows
        if not fast:
            if self.rng != "device":
                raise ValueError("realisation_spectra needs rng='device' (realisations addressed by index)")
            rows = one_by_one([first + i * stride for i in range(count)])
            self._realisation = first + count * stride
            return kc.copy(), np.array([r[1] for r in rows]).reshape(count, -1), np.array([r[2] for r in rows]).reshape(count, -1)
        self._set_amplitude(1. / (1. + redshift), linear)
        eng = self.engine
        eng.set_bins(bins, thr, amb)
        nb = bins.size
        nvox = float(self.N) ** 3
        shift = hostgeom.lognormal_shift(self._sigma2, nvox) if lognormal else 0.0
        R = eng.montecarlo_power(self.seed, first, count, stride=stride, pre_exp=lognormal, exp_shift=shift)
        self._realisation = first + count * stride
        self.last_realisation = self._realisation - stride if count else getattr(self, "last_realisation", None)
        s1, s2, esum = R[:, 0:2 * nb:2], R[:, 1:2 * nb:2], R[:, 2 * nb]
        cnt = eng.bin_counts()
        good = np.ones(count, dtype=bool)
        with np.errstate(all="ignore"):
            if lognormal:
                good = hostgeom.lognormal_sums_in_range_many(cnt, s1, s2, esum)
                mean = esum / nvox
                s1, s2 = hostgeom.lognormal_rescale(s1, s2, mean)
            vals, std = hostgeom.finish_bins_many(cnt, s1, s2, self.boxfactor, _eps_of(eng))
        for i in np.nonzero(~good)[0]:
            # a realisation whose extremes fall outside what the variance-based shift allows for: once more on its own, where
            # the step is repeated with the shift from its maximum (binned_power_spectrum)
            keep = self._realisation
            row = one_by_one([first + int(i) * stride])[0]
            self._realisation = keep
            vals[i], std[i] = row[1], row[2]
        return kc.copy(), vals, std

    # ----------------------------------------------------------- velocity / potential
    def _field_k(self, delta_x, delta_k):
        if delta_x is not None and delta_k is not None:
            raise ValueError("delta_x and delta_k specified; can only specify one")
        if delta_x is not None:
            return self.engine.fft_r2c(self._as_real(delta_x))
        if delta_k is None:
            return self.delta_k
        return self._as_spectrum(delta_k)

    def realise_velocity(self, delta_x=None, delta_k=None, redshift=None, inplace=True):
        """v(k) = i [f H a] delta_k k_vec / k^2 per component (box.py:197-290)."""
        if redshift is None:
            redshift = self.redshift
        scale_factor = 1. / (1. + redshift)
        if delta_x is not None and delta_k is not None:
            raise ValueError("delta_x and delta_k specified; can only specify one")
        if self.N % 2 != 0:
            raise UnboundLocalError("local variable 'mx' referenced before assignment")
        fac = 100. * self.cosmo['h'] * _ccl.h_over_h0(self.cosmo, a=scale_factor) \
            * _ccl.growth_rate(self.cosmo, a=scale_factor) * scale_factor
        src = getattr(self, "delta_x", None)
        if delta_x is None and delta_k is None and self._delta_k is None \
                and isinstance(src, PendingDensity) and src.generator is not None:
            # the stored realisation came from the counter-based generator: delta_k = fftn(delta_x) is only
            # formed if a component is read in k space; to_real() regenerates it inside its first FFT pass
            box, cache = self, {}

            def dk_of():
                if "dk" not in cache:
                    cache["dk"] = box._delta_k if box.delta_x is src and box._delta_k is not None \
                        else box.engine.fft_r2c(src)
                return cache["dk"]
            velocity_k = tuple(LazySpectrum(self.engine, HALF, lambda c=c: self.engine.velocity_k(dk_of(), c, fac),
                                            recipe=src.generator + (c, fac)) for c in range(3))
        else:
            dk = self._field_k(delta_x, delta_k)
            velocity_k = tuple(LazySpectrum(self.engine, dk.kind, lambda c=c: self.engine.velocity_k(dk, c, fac))
                               for c in range(3))    # each component is computed when first used
        if inplace:
            self.velocity_k = velocity_k
        return velocity_k

    def realise_potential(self, delta_x=None, delta_k=None, redshift=None, inplace=True):
        """delta_k / k^2 with the monopole zeroed; like the reference the physical
        prefactor is not applied (box.py:344-348)."""
        dk = self._field_k(delta_x, delta_k)
        phi_k = self.engine.potential_k(dk)
        if inplace:
            self.phi_k = phi_k
        return phi_k

    # ------------------------------------------------------------------- filters
    def _filter_table(self, transfer_fn, layout_kind):
        """Evaluate an arbitrary callable on the host grid (as box.py:374-378 does)."""
        N = self.N
        m = self._modes
        ones = np.ones((N, N, N))
        k_perp = 2. * np.pi * np.sqrt(((m / self.Lx) ** 2.)[:, None, None] + ((m / self.Ly) ** 2.)[None, :, None]) * ones
        k_par = (2. * np.pi * m / self.Lz)[None, None, :] * ones
        with np.errstate(all="ignore"):
            t = np.asarray(transfer_fn(k_perp, k_par))
        if np.iscomplexobj(t):
            raise TypeError("complex-valued transfer functions are not supported on the device")
        t = np.broadcast_to(t, (N, N, N)).astype(self.engine.rdtype)
        if layout_kind == HALF:
            padded = np.zeros((N, self.engine.rows, self.engine.pitch), dtype=self.engine.rdtype)
            padded[:, :N, :N // 2 + 1] = t[:, :, :N // 2 + 1]
            t = padded
        return self.engine.upload_raw(t)

    def apply_transfer_fn(self, field_k, transfer_fn):
        """ifftn(nan_to_num(field_k * T(k_perp, k_par))) (box.py:356-381); the result is
        complex like the reference's.  A Hermitian (device half-spectrum) field with a
        ``DeviceFilter`` that is even in k_par takes the real-field fast path."""
        eng = self.engine
        f = self._as_spectrum(field_k)
        if isinstance(transfer_fn, DeviceFilter):
            if f.kind == HALF and transfer_fn.even_in_kpar:
                return FilteredField(eng, f, transfer_fn)      # lazy: see the class
            if f.kind == HALF:
                f = eng.expand_half(f)
            dk = eng.apply_filter(f, transfer_fn.kind, transfer_fn.params)
        else:
            if f.kind == HALF:
                f = eng.expand_half(f)     # an arbitrary callable need not be even in k_par
            table = self._filter_table(transfer_fn, FULL)
            dk = eng.apply_filter(f, FB_FILT_TABLE, table=table)
        if dk.kind == HALF:
            return eng.fft_c2r(dk, destroy=True, as_complex=True)
        return eng.fft_c2c(dk, +1, 1.0 / self.N ** 3, inplace=True)

    apply_filter = apply_transfer_fn

    def window(self, k, R):
        """Top-hat window squared (box.py:595-613)."""
        return self.window1(k, R) ** 2.

    def window1(self, k, R):
        """Top-hat window (box.py:615-633)."""
        x = k * R
        return (3. / x ** 3.) * (np.sin(x) - x * np.cos(x))

    def smooth_field(self, field_k, R):
        """Top-hat smoothing with radius R Mpc/h (box.py:635-655)."""
        eng = self.engine
        f = self._as_spectrum(field_k)
        dk = eng.apply_filter(f, FB_FILT_TOPHAT, (R / self.cosmo['h'], 0, 0, 0))
        if dk.kind == HALF:
            return eng.fft_c2r(dk, destroy=True, as_complex=True)
        return eng.fft_c2c(dk, +1, 1.0 / self.N ** 3, inplace=True)

    # ---------------------------------------------------------------- redshift space
    def redshift_space_density(self, delta_x=None, velocity_z=None, sigma_nl=0., method='linear'):
        """Line-of-sight remap to redshift space (box.py:384-438).  ``method`` is what the reference hands to scipy's
        ``griddata`` (box.py:433-437), all three of its one-dimensional rules on the device: 'linear' (default), 'nearest', and
        'cubic' -- the not-a-knot cubic spline through every line of sight's sorted shifted samples (what
        ``griddata(method='cubic')`` = ``interp1d(kind='cubic')`` is in one dimension), the fill value outside them."""
        if method not in self.engine.RSD_METHODS:
            raise ValueError("Unknown interpolation method %r for 1 dimensional data" % (method,))      # scipy's own refusal
        Hz = 100. * self.cosmo['h'] * _ccl.h_over_h0(self.cosmo, self.scale_factor)
        d = self._as_real(delta_x)
        v = self._as_real(velocity_z)
        noise = None
        if sigma_nl > 0. and self.rng == "numpy":
            # the reference draws N normals per line of sight in (i, j) order (box.py:416-418)
            noise = self.engine.upload(np.random.normal(0., 1., (self.N, self.N, self.N)), REAL)
        seed = self.seed + 0x9E3779B97F4A7C15 * (self._realisation + 1)
        if noise is None and method != "cubic" and self.engine.fuses_redshift_space:
            lazy = RedshiftSpaceField(self.engine, d, v, (Hz, sigma_nl, seed, method))
            if lazy.fusable():
                return lazy
        return self.engine.redshift_space(d, v, Hz, sigma_nl, noise, seed, method)

    # ------------------------------------------------------------------- log-normal
    def lognormal(self, delta_x):
        """exp(delta)/<exp(delta)> - 1 (box.py:441-460).  Returned lazily: the field is
        computed when it is first read, and ``binned_power_spectrum(delta_x=...)`` of it
        fuses the exp() into the first FFT pass instead of materialising it."""
        return LognormalField(self.engine, self._as_real(delta_x))

    # ----------------------------------------------------------------- power spectrum
    def _shell_thresholds(self, bins):
        """Bin as a step function of the integer shell n^2 (cubic boxes).  A shell whose
        |k| is within rounding of an edge is left to the exact on-device expression."""
        N = self.N
        n2 = np.arange(3 * (N // 2) ** 2 + 1, dtype=np.float64)
        k = 2. * np.pi * np.sqrt(n2) / self.Lx
        eps = 64 * np.finfo(np.float64).eps
        lo = np.digitize(k * (1. - eps), bins)
        hi = np.digitize(k * (1. + eps), bins)
        amb = np.nonzero(lo != hi)[0]
        if amb.size > 8:
            return None, ()
        thr = np.searchsorted(hi, np.arange(1, bins.size + 1), side="left")
        return thr.astype(np.int32), tuple(int(a) for a in amb)

    def _bin_setup(self, nbins, kbins):
        """Edges, centres of bins 1..nbins-1 (box.py:745-751) and the shell thresholds, remembered per
        bin set: a Monte-Carlo loop asks for the same bins every realisation."""
        key = ("n", int(nbins)) if kbins is None else ("k", np.asarray(kbins, dtype=np.float64).tobytes())
        hit = self._bin_cache.get(key)
        if hit is not None:
            return hit
        if kbins is not None:
            bins = np.array(kbins, dtype=np.float64)
        else:
            bins = np.logspace(np.log10(self.kmin), np.log10(self.kmax), nbins)   # box.py:749
        _bins = [0.0] + list(bins)
        cent = [0.5 * (_bins[j + 1] + _bins[j]) for j in range(bins.size)]
        kc = np.array(cent[1:])
        thr, amb = (None, ())
        if self._cubic and np.all(np.diff(bins) >= 0):
            thr, amb = self._shell_thresholds(bins)
        bins.setflags(write=False)
        if len(self._bin_cache) > 16:
            self._bin_cache.clear()
        self._bin_cache[key] = (bins, kc, thr, amb)
        return self._bin_cache[key]

    def binned_power_spectrum(self, delta_x=None, delta_k=None, nbins=20, kbins=None, wait=True, keep_field=True):
        """Shell-averaged power spectrum of the realisation (box.py:696-768): bin centres,
        mean of |delta_k|^2/boxfactor and std/sqrt(n) per bin; bin 0 is dropped and empty
        bins are NaN, as in the reference.  ``wait=False`` (additive) returns a
        ``PendingSpectrum`` at once; its ``result()`` gives the same triple later, so that
        many realisations can be queued without a host round trip each.
        ``keep_field=False`` (additive; device-generator realisations whose last pass is still pending): the fused
        z pass does not write delta_x -- a Monte-Carlo loop that only wants spectra saves an eighth of the step's
        traffic; reading the field afterwards draws the same realisation again (same seed, same index)."""
        if delta_x is not None and delta_k is not None:
            raise ValueError("delta_x and delta_k specified; can only specify one")
        bins, kc, thr, amb = self._bin_setup(nbins, kbins)
        kc = kc.copy()                    # callers own what they get back
        eng = self.engine
        eng.set_bins(bins, thr, amb)

        if isinstance(delta_x, FilteredField):
            # P(k) of apply_transfer_fn's result = shell sums of |field_k T|^2 (Hermitian field, even filter)
            root, src = delta_x._root(), delta_x.spectrum
            filt = (delta_x.filter.kind, delta_x.filter.params)
            if root._x_done is None and not root.materialised and thr is not None \
                    and isinstance(src, LazySpectrum) and src.source_real is not None and not src.materialised:
                # what apply_transfer_fn returns is the field (box.py:381): the pass that filters and bins an x line
                # also transforms it back, and reading the field later costs the y and z passes only
                rs = src.source_real
                if isinstance(rs, RedshiftSpaceField) and rs.fusable():
                    res, root._x_done = rs.consume(filt, field=True)       # the z passes of the whole chain in one kernel
                else:
                    res, root._x_done = eng.power_filtered(rs, filt, field=True)
                pending = PendingSpectrum(eng, res, bins.size, kc, self.boxfactor, None, None)
                return pending if not wait else pending.result()
            cnt, s1, s2 = eng.bin_power(src, filt=filt)
            out = (kc,) + _finish_bins(cnt, s1, s2, self.boxfactor, _eps_of(eng))
            return out if wait else _Ready(out)

        if delta_x is None and delta_k is None and thr is not None and self._delta_k is None \
                and getattr(self, "delta_x", None) is not None:
            # the stored realisation, whose fftn has not been asked for yet (box.py:736-739 would use
            # self.delta_k = fftn(delta_x)): same numbers through the fused path below, no spectrum stored
            delta_x = self.delta_x
        if delta_x is not None and thr is not None:
            # fused path (cubic boxes): r2c with the binning inside the last pass
            ln = isinstance(delta_x, LognormalField) and not delta_x.materialised and bins[0] > 0.
            src = delta_x.source if ln else self._as_real(delta_x)
            # log-normal: exp(d - shift) instead of exp(d) -- the estimate exp(d)/<exp(d)> - 1 is the same, and with the
            # right shift a single-precision plan's sums stay inside the float range (a non-linear P(k) sampled at
            # 2 Mpc gives sigma = 8, exp(d) reaches 1e19 and |delta_k|^4 would overflow; at 0.5 Mpc sigma = 21).  A field
            # this box drew carries the variance of its distribution: the shift comes from that without a look at the
            # data (hostgeom.lognormal_shift; a realisation whose extremes fall outside is repeated with the exact shift,
            # PendingSpectrum.result).  Any other field is asked for its maximum first.
            shift, redo = 0.0, None
            if ln:
                nvox = float(self.N) ** 3
                sigma2 = getattr(src, "sigma2", None)
                if sigma2 is not None:
                    shift = hostgeom.lognormal_shift(sigma2, nvox)
                elif eng.precision == "f64":
                    shift = 0.0       # double precision needs no shift until exp(d) itself leaves the range (where the reference's
                                      # float64 does too), and a look at the data would wait for the stream (wait=False pipelines)
                else:
                    shift = hostgeom.lognormal_shift_exact(eng.max_real(src), nvox)
                nb_, bin_args = bins.size, (bins, thr, amb)
                # (the repeat must not keep the field alive: a loop that queues hundreds of spectra drops each delta_x
                # at once, and its buffer goes back to the pool; a field that is gone is drawn again from its recipe)
                wsrc, regen = weakref.ref(src), getattr(src, "_regenerate", None)

                def redo():
                    field = wsrc()
                    if field is None:
                        if regen is None:
                            raise FloatingPointError("log-normal P(k): the exponentials left the plan's floating-point "
                                                     "range and the field is no longer there to repeat the step")
                        field = regen()
                    self.lognormal_repeats += 1
                    eng.set_bins(*bin_args)
                    exact = hostgeom.lognormal_shift_exact(eng.max_real(field), nvox)   # (materialises a pending field)
                    res2, _ = eng.power_fused(field, pre_exp=True, exp_shift=exact)
                    return eng.fetch_results(res2, nb_)
            if isinstance(src, RedshiftSpaceField) and src.fusable() and not ln:
                res, _ = src.consume()
            elif isinstance(src, PendingDensity) and not src.materialised and src._pending is not None:
                res, real = eng.power_pending(src._pending, pre_exp=ln, exp_shift=shift, keep_field=keep_field)     # z passes fused
                if real is not None:
                    src._adopt(real)
                else:
                    src._pending = None        # consumed; reading the field later regenerates it
            else:
                res, _ = eng.power_fused(src, pre_exp=ln, exp_shift=shift)
            pending = PendingSpectrum(eng, res, bins.size, kc, self.boxfactor, self.N ** 3 if ln else None, None, redo)
            return pending if not wait else pending.result()

        if delta_x is not None:
            spec = eng.fft_r2c(self._as_real(delta_x))
        elif delta_k is None:
            spec = self.delta_k
        else:
            spec = self._as_spectrum(delta_k)
        cnt, s1, s2 = eng.bin_power(spec)
        out = (kc,) + _finish_bins(cnt, s1, s2, self.boxfactor, _eps_of(eng))
        return out if wait else _Ready(out)

    def sigmaR(self, R):
        """RMS of the field smoothed with a top-hat of R Mpc/h, from the binned power
        spectrum (box.py:657-683; scipy's simps is spelled simpson since 1.14)."""
        k, pk, stddev = self.binned_power_spectrum()
        good = ~np.isnan(pk)
        pk, k = pk[good], k[good]
        y = k ** 2. * pk * self.window(k, R / self.cosmo['h'])
        I = _simpson(y, x=k)
        return np.sqrt(I / (2. * np.pi ** 2.))

    def sigma8(self):
        return self.sigmaR(8.0)

    def theoretical_power_spectrum(self):
        """box.py:770-782."""
        k = np.logspace(-3.5, 1., int(1e3))
        pk = _ccl.nonlin_matter_power(self.cosmo, k=k, a=self.scale_factor)
        return k, pk

    # ----------------------------------------------------------------- coordinates
    def freq_array(self, redshift=None):
        """Channel frequencies in MHz, decreasing along z (box.py:789-828)."""
        if redshift is None:
            redshift = self.redshift
        a = 1. / (1. + redshift)
        dx = self.Lz / self.N
        Hz = 100. * self.cosmo['h'] * _ccl.h_over_h0(self.cosmo, a)
        df = dx * self.line_freq * (a ** 2. * Hz) / (C / 1e3)
        freqs = a * self.line_freq + df * (np.arange(self.N) - 0.5 * (self.N - 1.))
        return freqs[::-1]

    def pixel_array(self, redshift=None):
        """Angular pixel coordinates in degrees (box.py:831-864)."""
        if redshift is None:
            redshift = self.redshift
        r = _ccl.comoving_angular_distance(self.cosmo, 1. / (1. + redshift))
        ang_x = (180. / np.pi) * ((self.x[1] - self.x[0]) / r)
        ang_y = (180. / np.pi) * ((self.y[1] - self.y[0]) / r)
        grid = np.arange(self.N) - 0.5 * (self.N - 1.)
        return ang_x * grid, ang_y * grid

    # ------------------------------------------------------------------ self tests
    def test_parseval(self):
        """sum(delta_x^2) N^3 against sum |delta_k|^2 (box.py:931-948), both reduced on
        the device in fp64."""
        s1 = self.engine.sum_real(self.delta_x, squared=True) * self.N ** 3.
        s2 = self.engine.sumsq_half(self.delta_k) if self.delta_k.kind == HALF \
            else float(np.sum(np.abs(self.delta_k.host()) ** 2))
        print("Parseval test:", s1 / s2, "(should be 1.0)")
        return s1, s2

    def test_sampling_error(self):
        """sigma8 of the realisation against theory (box.py:871-928)."""
        h = self.cosmo['h']
        s8_real = self.sigma8()
        _k = np.linspace(self.kmin, self.kmax, int(5e3))
        _pk = _ccl.nonlin_matter_power(self.cosmo, k=_k, a=self.scale_factor)
        _y = np.nan_to_num(_k ** 2. * _pk * self.window(_k, 8.0 / h))
        s8_th_win = np.sqrt(_simpson(_y, x=_k) / (2. * np.pi ** 2.))
        _k2 = np.logspace(-5, 2, int(5e4))
        _pk2 = _ccl.nonlin_matter_power(self.cosmo, k=_k2, a=self.scale_factor)
        _y2 = np.nan_to_num(_k2 ** 2. * _pk2 * self.window(_k2, 8.0 / h))
        s8_th_full = np.sqrt(_simpson(_y2, x=_k2) / (2. * np.pi ** 2.))
        s8_realspace = np.std(np.asarray(self.smooth_field(self.delta_k, 8.0)))
        s20_realspace = np.std(np.asarray(self.smooth_field(self.delta_k, 20.0)))
        s20_real = self.sigmaR(20.)
        print("")
        print("sigma8 (real.): \t", s8_real)
        print("sigma8 (th.win.):\t", s8_th_win)
        print("sigma8 (th.full):\t", s8_th_full)
        print("sigma8 (realsp.):\t", s8_realspace)
        print("ratio =", 1. / (s8_real / s8_realspace))
        print("")
        print("sigma20 (real.): \t", s20_real)
        print("sigma20 (realsp.):\t", s20_realspace)
        print("ratio =", 1. / (s20_real / s20_realspace))
        print("var(delta) =", np.sqrt(self.engine.sum_real(self.delta_x, True) / self.N ** 3
                                      - (self.engine.sum_real(self.delta_x) / self.N ** 3) ** 2))
