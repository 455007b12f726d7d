"""
TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float64) of the reference's
3-D density-field hot path, `fastbox/box.py` of philbull/FastBox v0.0.9.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module, and only as the checker / timed baseline.
The product (`fastbox_amd/`) never imports it and has no CPU fallback.

Pinning: `tests/test_oracle.py` checks every function here (a) bit-for-bit
against the reference itself, loaded through `oracle/ref_loader.py`, when
/root/reference is present (build container), and (b) against the golden
vectors in `tests/golden/*.npz`, which `oracle/make_golden.py` captured from
the reference run in the build container (same numpy 2.2 / scipy 1.15).
The one third-party input, pyccl's P(k)/E/f/D, is "parity unpinned"
(SURVEY.md 8c): it enters here only as callables/scalars supplied by the
caller.

Each function cites the reference lines it restates.  The arithmetic is kept
in the reference's order of operations so that float64 results are identical,
not merely close.
"""
import contextlib
import numpy as np
from numpy import fft as _fft


@contextlib.contextmanager
def threaded_fft(workers):
    """bench.py's best-effort CPU line: the same functions with scipy.fft on `workers` threads in place of
    numpy's single-threaded pocketfft.  Results agree to rounding, not bit for bit: never used by a parity test."""
    global _fft
    import scipy.fft as sfft

    class _Threaded(object):
        fftfreq = staticmethod(np.fft.fftfreq)

        @staticmethod
        def fftn(a):
            return sfft.fftn(a, workers=workers)

        @staticmethod
        def ifftn(a):
            return sfft.ifftn(a, workers=workers)
    keep, _fft = _fft, _Threaded
    try:
        yield _Threaded
    finally:
        _fft = keep


# --------------------------------------------------------------------------
# geometry                                                   box.py:75-101
# --------------------------------------------------------------------------
def box_geometry(box_scale, nsamp):
    """Grid coordinates, side lengths, DFT volume factor and k range.

    box.py:76-89 (linspace grid *including* both end points, so the side is
    x[-1]-x[0]), :94 (boxfactor = N^6 / (Lx Ly Lz)), :100-101 (kmin, kmax;
    kmax uses N, not N/2).
    """
    if isinstance(box_scale, tuple):
        assert len(box_scale) == 3
        axes = [np.linspace(-0.5 * s, 0.5 * s, nsamp) for s in box_scale]
    else:
        ax = np.linspace(-0.5 * box_scale, 0.5 * box_scale, nsamp)
        axes = [ax, ax, ax]
    L = [a[-1] - a[0] for a in axes]
    g = dict(N=nsamp, x=axes[0], y=axes[1], z=axes[2],
             Lx=L[0], Ly=L[1], Lz=L[2])
    g['boxfactor'] = (nsamp ** 6.) / (L[0] * L[1] * L[2])
    g['kmin'] = 2. * np.pi / np.max(L)
    g['kmax'] = 2. * np.pi * np.sqrt(3.) * nsamp / np.min(L)
    return g


def mode_numbers(N):
    """Mode number held at each grid index, box.py:116-123.  The reference uses the
    truncated values NN = (N*fftfreq(N)).astype('i') as *indices* (`Kx[i,:,:] = i`), so
    for sizes where N*(j/N) rounds below an integer (e.g. N = 24) some indices are
    never assigned and stay 0; that behaviour is reproduced, not repaired."""
    NN = (N * _fft.fftfreq(N, 1.)).astype("i")
    K = np.zeros(N)
    for i in NN:
        K[i] = i
    return K


def k_magnitude(g):
    """|k| on the full (N,N,N) grid, box.py:125-127, via broadcasting rather
    than three materialised integer grids (same per-element operations)."""
    m = mode_numbers(g['N'])
    ax = (m / g['Lx']) ** 2.
    ay = (m / g['Ly']) ** 2.
    az = (m / g['Lz']) ** 2.
    return 2. * np.pi * np.sqrt(ax[:, None, None] + ay[None, :, None]
                                + az[None, None, :])


def k_perp_par(g):
    """k_perp (N,N,1) and signed k_par (1,1,N), box.py:374-375."""
    m = mode_numbers(g['N'])
    kperp = 2. * np.pi * np.sqrt(((m / g['Lx']) ** 2.)[:, None, None]
                                 + ((m / g['Ly']) ** 2.)[None, :, None])
    kpar = (2. * np.pi * m / g['Lz'])[None, None, :]
    return kperp, kpar


# --------------------------------------------------------------------------
# Gaussian realisation                                       box.py:130-194
# --------------------------------------------------------------------------
def coloured_noise(g, pk_fn, re, im):
    """X(k) = (re + i im) sqrt(nan_to_num(P(k)) boxfactor), box.py:161-176."""
    k = k_magnitude(g)
    pk = np.reshape(pk_fn(k.flatten()), k.shape)
    pk = np.nan_to_num(pk)
    pk *= g['boxfactor']
    return (re + 1j * im) * np.sqrt(pk)


def draw_noise(N, rng=np.random):
    """`re` then `im`, each (N,N,N) C order, from the legacy global stream
    (box.py:174-175)."""
    re = rng.normal(0.0, 1.0, (N, N, N))
    im = rng.normal(0.0, 1.0, (N, N, N))
    return re, im


def realise_density(g, pk_fn, re, im):
    """delta_x = Re ifftn(X); delta_k = fftn(delta_x).  box.py:187-193."""
    X = coloured_noise(g, pk_fn, re, im)
    delta_x = _fft.ifftn(X).real
    delta_k = _fft.fftn(delta_x)
    return delta_x, delta_k


# --------------------------------------------------------------------------
# shell-binned power spectrum                                box.py:696-768
# --------------------------------------------------------------------------
def default_kbins(g, nbins):
    """box.py:749."""
    return np.logspace(np.log10(g['kmin']), np.log10(g['kmax']), nbins)


def binned_power_spectrum(g, delta_k, nbins=20, kbins=None):
    """Centres, per-bin mean of |delta_k|^2/boxfactor and std/sqrt(n).

    box.py:741-768: bins are *edges*; np.digitize index i means
    bins[i-1] <= k < bins[i]; only indices 0..nbins-1 are averaged and index 0
    is dropped; empty bins give NaN; population std (ddof=0).
    """
    pk = delta_k * np.conj(delta_k)
    pk = pk.real / g['boxfactor']
    bins = np.asarray(kbins) if kbins is not None else default_kbins(g, nbins)
    _b = [0.0] + list(bins)
    cent = [0.5 * (_b[j + 1] + _b[j]) for j in range(bins.size)]
    idxs = np.digitize(k_magnitude(g).flatten(), bins)
    flat = pk.flatten()
    vals = np.zeros(bins.size)
    err = np.zeros(bins.size)
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i in range(bins.size):
                sel = flat[idxs == i]
                vals[i] = np.mean(sel)
                err[i] = np.std(sel) / np.sqrt(sel.size)
    return np.array(cent[1:]), vals[1:], err[1:]


# --------------------------------------------------------------------------
# transfer function / smoothing                   box.py:356-381, 635-655
# --------------------------------------------------------------------------
def apply_transfer_fn(g, field_k, transfer_fn):
    """ifftn(nan_to_num(field_k T(k_perp,k_par))) -- complex result."""
    kperp, kpar = k_perp_par(g)
    full = np.broadcast_to(kperp, field_k.shape), np.broadcast_to(kpar, field_k.shape)
    dk = field_k * transfer_fn(full[0], full[1])
    dk = np.nan_to_num(dk)
    return _fft.ifftn(dk)


def tophat_window1(k, R):
    """box.py:631-633."""
    x = k * R
    with np.errstate(all='ignore'):
        return (3. / x ** 3.) * (np.sin(x) - x * np.cos(x))


def tophat_window(k, R):
    """box.py:595-612: the window squared."""
    return tophat_window1(k, R) ** 2.


def sigma_R(g, delta_k, R, h):
    """box.py:657-683: sqrt(int k^2 P(k) W^2(k R/h) dk / 2 pi^2) by Simpson's rule over the non-NaN bins of the
    default binned power spectrum (scipy.integrate.simps of the reference's day = simpson)."""
    from scipy.integrate import simpson
    k, pk, stddev = binned_power_spectrum(g, delta_k)
    good = ~np.isnan(pk)
    pk, k = pk[good], k[good]
    y = k ** 2. * pk * tophat_window(k, R / h)
    return np.sqrt(simpson(y, x=k) / (2. * np.pi ** 2.))


def sampling_report(g, delta_x, delta_k, pk_fn, h):
    """The nine numbers box.py:871-928 (test_sampling_error) prints, in print order: sigma8 from the realisation,
    theory in the box's k-window, theory over a wide window, real-space sigma8, their ratio, the same three for
    R = 20 Mpc/h, std(delta_x)."""
    from scipy.integrate import simpson
    s8_real = sigma_R(g, delta_k, 8., h)
    _k = np.linspace(g['kmin'], g['kmax'], int(5e3))
    _y = np.nan_to_num(_k ** 2. * pk_fn(_k) * tophat_window(_k, 8.0 / h))
    s8_th_win = np.sqrt(simpson(_y, x=_k) / (2. * np.pi ** 2.))
    _k2 = np.logspace(-5, 2, int(5e4))
    _y2 = np.nan_to_num(_k2 ** 2. * pk_fn(_k2) * tophat_window(_k2, 8.0 / h))
    s8_th_full = np.sqrt(simpson(_y2, x=_k2) / (2. * np.pi ** 2.))
    s8_realspace = np.std(smooth_field(g, delta_k, 8.0, h))
    s20_realspace = np.std(smooth_field(g, delta_k, 20.0, h))
    s20_real = sigma_R(g, delta_k, 20., h)
    return np.array([s8_real, s8_th_win, s8_th_full, s8_realspace, 1. / (s8_real / s8_realspace),
                     s20_real, s20_realspace, 1. / (s20_real / s20_realspace), np.std(delta_x)])


def smooth_field(g, field_k, R, h):
    """box.py:651-655 (R in Mpc/h)."""
    dk = field_k * tophat_window1(k_magnitude(g), R / h)
    dk = np.nan_to_num(dk)
    return _fft.ifftn(dk)


# --------------------------------------------------------------------------
# log-normal                                                 box.py:457-460
# --------------------------------------------------------------------------
def lognormal(delta_x):
    d = np.exp(delta_x)
    d /= np.mean(d)
    d -= 1.
    return d


# --------------------------------------------------------------------------
# velocity / potential                            box.py:251-285, 347-348
# --------------------------------------------------------------------------
def realise_velocity(g, delta_k, fac):
    """(v_x, v_y, v_z)(k) = i fac delta_k k_j / k^2; NaN -> 0; for even N the
    most negative mode plane of each component's own axis is zeroed."""
    N = g['N']
    if N % 2 != 0:
        # the reference raises UnboundLocalError here (box.py:268-274)
        raise UnboundLocalError("reference realise_velocity is undefined for odd N")
    m = mode_numbers(N)
    k2 = k_magnitude(g) ** 2.
    Kx, Ky, Kz = m[:, None, None], m[None, :, None], m[None, None, :]
    with np.errstate(all='ignore'):
        Ax = 1.j * delta_k * Kx * (2. * np.pi / g['Lx']) / k2
        Ay = 1.j * delta_k * Ky * (2. * np.pi / g['Ly']) / k2
        Az = 1.j * delta_k * Kz * (2. * np.pi / g['Lz']) / k2
    Ax, Ay, Az = np.nan_to_num(Ax), np.nan_to_num(Ay), np.nan_to_num(Az)
    neg = int(np.argmin(m))
    Ax[neg, :, :] = 0.
    Ay[:, neg, :] = 0.
    Az[:, :, neg] = 0.
    Ax *= fac
    Ay *= fac
    Az *= fac
    return Ax, Ay, Az


def realise_potential(g, delta_k):
    """delta_k / k^2 with the monopole zeroed (prefactor is computed but not
    applied by the reference, box.py:344-348)."""
    with np.errstate(all='ignore'):
        phi = delta_k / k_magnitude(g) ** 2.
    phi[0, 0, 0] = 0.
    return phi


# --------------------------------------------------------------------------
# redshift-space remap                                       box.py:405-438
# --------------------------------------------------------------------------
def _regrid_linear_1d(s, vals, zgrid, fill):
    """What scipy.interpolate.griddata does for 1-D 'linear' (scipy 1.15
    _ndgriddata.py:303-317 -> interp1d -> np.interp, then out-of-range fill):
    sort the scattered abscissae, piecewise-linear interpolation
    slope*(x-x_lo)+y_lo with the left bracket found by bisection, values
    outside [min s, max s] replaced by `fill`."""
    order = np.argsort(s)
    order = order[np.argsort(s[order], kind="mergesort")]
    xs, ys = s[order], vals[order]
    out = np.interp(zgrid, xs, ys)
    out[zgrid < xs[0]] = fill
    out[zgrid > xs[-1]] = fill
    return out


def _regrid_nearest_1d(s, values, xi):
    """griddata(points, values, xi, method='nearest') for 1-D points (scipy 1.15 _ndgriddata.py:303-317: argsort, then
    interp1d(kind='nearest', fill_value='extrapolate'); _interpolate.py: x_bds = x/2, x_bds[1:] + x_bds[:-1];
    searchsorted(x_bds, x_new, side='left'), clipped): the sample whose neighbourhood between midpoints holds x_new,
    the lower one when x_new is a midpoint; beyond the end samples, the end samples (no fill value)."""
    idx = np.argsort(s)
    x, y = s[idx], values[idx]
    x_bds = x / 2.0
    x_bds = x_bds[1:] + x_bds[:-1]
    k = np.searchsorted(x_bds, xi, side='left').clip(0, len(x) - 1)
    return y[k]


def _regrid_cubic_1d(s, values, z, fill):
    """griddata(points=(s,), values, xi=(z,), method='cubic', fill_value=fill) in one dimension (box.py:433-437; scipy
    interpolate/_ndgriddata.py: argsort, then interp1d(kind='cubic', bounds_error=False, fill_value=fill)): the not-a-knot
    cubic spline through the sorted samples, `fill` outside [min s, max s].  Restated with CubicSpline(bc_type='not-a-knot')
    -- the same spline by another of scipy's routes (they agree to 1e-15 here; pinned against the reference's own call in
    tests/golden: rsd0_cubic)."""
    from scipy.interpolate import CubicSpline
    idx = np.argsort(s)
    cs = CubicSpline(s[idx], values[idx], bc_type='not-a-knot', extrapolate=False)
    y = cs(z)
    y[np.isnan(y)] = fill
    return y


def redshift_space_density(g, delta_x, velocity_z, Hz, sigma_nl=0.,
                           rng=np.random, method='linear'):
    """Per line of sight (i,j): s = z - (v_z + sigma_nl n)/H, periodic wrap,
    re-grid delta(s) on z; endpoint-average fill.  Noise is drawn LOS by LOS in
    (i,j) order from the legacy global stream (box.py:412-418).  method: 'linear', 'nearest' or 'cubic' (box.py:433-437)."""
    z = g['z']
    out = np.zeros_like(delta_x) - 1.
    zmin = np.min(z)
    length_z = np.max(z) - zmin
    for i in range(delta_x.shape[0]):
        for j in range(delta_x.shape[1]):
            vel_nl = 0.
            if sigma_nl > 0.:
                vel_nl = sigma_nl * rng.normal(0., 1., z.size)
            s = z - (velocity_z[i, j, :] + vel_nl) / Hz
            s = (s - zmin) % length_z + zmin
            fill = 0.5 * (delta_x[i, j, 0] + delta_x[i, j, -1])
            if method == 'nearest':
                out[i, j, :] = _regrid_nearest_1d(s, delta_x[i, j, :], z)
            elif method == 'cubic':
                out[i, j, :] = _regrid_cubic_1d(s, delta_x[i, j, :], z, fill)
            else:
                out[i, j, :] = _regrid_linear_1d(s, delta_x[i, j, :], z, fill)
    return out


# --------------------------------------------------------------------------
# consistency checks                                         box.py:931-948
# --------------------------------------------------------------------------
def parseval(delta_x, delta_k):
    N = delta_x.shape[0]
    s1 = np.sum(delta_x ** 2.) * N ** 3.
    s2 = np.sum(delta_k * np.conj(delta_k)).real
    return s1, s2


# --------------------------------------------------------------------------
# coordinates                                                box.py:789-864
# --------------------------------------------------------------------------
def freq_array(g, a, line_freq, Hz):
    """Hz = 100 h E(a) in km/s/Mpc.  box.py:813-828."""
    C = 299792458.
    dx = g['Lz'] / g['N']
    df = dx * line_freq * (a ** 2. * Hz) / (C / 1e3)
    freqs = a * line_freq + df * (np.arange(g['N']) - 0.5 * (g['N'] - 1.))
    return freqs[::-1]


def pixel_array(g, r):
    """r = comoving distance to the box centre.  box.py:854-864."""
    ang_x = (180. / np.pi) * ((g['x'][1] - g['x'][0]) / r)
    ang_y = (180. / np.pi) * ((g['y'][1] - g['y'][0]) / r)
    grid = np.arange(g['N']) - 0.5 * (g['N'] - 1.)
    return ang_x * grid, ang_y * grid
