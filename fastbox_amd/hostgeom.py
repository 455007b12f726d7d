"""
Host-side geometry of a CosmoBox that needs no GPU: grid, |k| shells, bin edges and the
per-bin shell thresholds handed to the device.  Every expression follows the reference's
numpy arithmetic (fastbox/box.py lines cited) so that |k| and np.digitize agree bit for bit.
Shared by ``CosmoBox`` (single GPU) and ``SlabBox`` (one box over several GPUs).
"""
import numpy as np


def grid(box_scale, nsamp):
    """x, y, z (linspace incl. both end points), side lengths, boxfactor, kmin, kmax
    (box.py:76-101)."""
    if isinstance(box_scale, tuple):
        assert len(box_scale) == 3, "Must specify scale of x, y, z dimensions"
        x, y, z = [np.linspace(-0.5 * s, 0.5 * s, nsamp) for s in box_scale]
    else:
        x = y = z = np.linspace(-0.5 * box_scale, 0.5 * box_scale, nsamp)
    L = (x[-1] - x[0], y[-1] - y[0], z[-1] - z[0])
    boxfactor = (nsamp ** 6.) / (L[0] * L[1] * L[2])
    kmin = 2. * np.pi / np.max(L)
    kmax = 2. * np.pi * np.sqrt(3.) * nsamp / np.min(L)
    return dict(N=nsamp, x=x, y=y, z=z, L=L, boxfactor=boxfactor, kmin=kmin, kmax=kmax,
                cubic=(L[0] == L[1] == L[2]))


def mode_numbers(N):
    """The integer mode number the reference's k grid carries at every index (box.py:119-123):

        NN = (N * fftfreq(N, 1.)).astype("i");  for i in NN: Kx[i, :, :] = i

    -- `i` is index AND value.  For a power of two the product is exact and this is fftfreq's numbering.  For other N the
    product can round just below an integer (24 * (7/24) = 6.999...), the cast truncates, NN then holds 6 twice and 7 never,
    and Kx[7] keeps the 0 it was initialised with.  Reproduced as it is: same inputs, same results as the reference."""
    nn = (N * np.fft.fftfreq(N, 1.)).astype("i")
    m = np.zeros(N, dtype=np.float64)
    for i in nn:
        m[i] = i
    return m


def axis_tables(N, L):
    """(m/L)**2 per axis [3N], m*(2 pi/L) per axis [3N], 2 pi m / Lz [N]
    (box.py:125-127, 254-256, 375)."""
    m = mode_numbers(N)
    axis2 = np.concatenate([(m / l) ** 2. for l in L])
    ksc = np.concatenate([m * (2. * np.pi / l) for l in L])
    kpar = 2. * np.pi * m / L[2]
    return axis2, ksc, kpar


def shell_wavenumbers(N, Lside):
    """|k| of every integer shell n^2 = i^2 + j^2 + l^2 of a cubic box."""
    n2 = np.arange(3 * (N // 2) ** 2 + 1, dtype=np.float64)
    return 2. * np.pi * np.sqrt(n2) / Lside


def shell_amplitude(N, Lside, boxfactor, pk_fn):
    """sqrt(nan_to_num(P(k)) * boxfactor) per shell (box.py:161-171)."""
    with np.errstate(all="ignore"):
        pk = np.nan_to_num(np.asarray(pk_fn(shell_wavenumbers(N, Lside)), dtype=np.float64))
        return np.sqrt(pk * boxfactor)


def shell_multiplicity(N):
    """Number of modes (i, j, l) of the full (N, N, N) grid on every integer shell n^2 = i^2 + j^2 + l^2."""
    m2 = (mode_numbers(N) ** 2).astype(np.int64)
    h2 = np.bincount((m2[:, None] + m2[None, :]).ravel())            # pairs (i, j)
    out = np.zeros(3 * (N // 2) ** 2 + 1, dtype=np.float64)
    vals, cnt = np.unique(m2, return_counts=True)
    for v, c in zip(vals, cnt):
        out[v:v + h2.size] += c * h2
    return out


def field_variance_cubic(N, amp_shells):
    """Variance of the realised field whose modes have E|delta_k|^2 = amp^2: sum_k amp_k^2 / N^6 (Parseval)."""
    a = np.asarray(amp_shells, dtype=np.float64)
    return float(np.sum(shell_multiplicity(N)[:a.size] * a * a) / float(N) ** 6)


def field_variance_sym(N, amp_sym):
    """The same from the (|m_x|, |m_y|, |m_z|) table of a box of any shape (N/2+1 entries per axis)."""
    w = np.full(N // 2 + 1, 2.0)
    w[0] = w[-1] = 1.0
    a = np.asarray(amp_sym, dtype=np.float64)
    return float(np.einsum("i,j,k,ijk->", w, w, w, a * a) / float(N) ** 6)


LN_SUM_MAX = 21.0        # ln of the largest sum of shifted exponentials a single-precision plan can carry: the k = 0
                         # mode of the transform IS that sum and its |.|^4 must stay below 2^128 (e^88.7)


def lognormal_shift(sigma2, nvox):
    """Shift c of the fused log-normal transform, which forms exp(delta - c): the estimate exp(d)/<exp(d)> - 1
    (box.py:457-460) and its P(k) do not depend on c, but on a single-precision plan the sums only stay in range
    for the right c.  Chosen from where S = sum exp(delta) of nvox Gaussian values of variance sigma2 will lie:

      sigma <= a = sqrt(2 ln nvox):  the sum is the sample mean's,  ln S = ln nvox + sigma^2/2  ->  S e^-c = e^7
      sigma >  a                  :  the sum is its few largest terms, ln S ~ max delta, a Gumbel variable of
                                     location mu = sigma (a - (ln ln nvox + ln 4 pi) / (2 a)) and scale sigma / a
                                     ->  the largest term lands at e^-6 (27 e-folds of head room above, since the
                                     upper tail of the maximum is the wide one; 14 below before anything that
                                     matters flushes to zero)

    (sigma^2/2, round 2's choice, is the first line without the ln nvox; past sigma ~ 13 it puts EVERY term below
    the single-precision range: 2048^3 at 2 Mpc/voxel has sigma = 21 and max delta - sigma^2/2 = -94.)
    A realisation that falls outside the range anyway is detected by its non-finite sums and repeated with the
    exact shift, `lognormal_shift_exact`."""
    sigma2 = max(float(sigma2), 0.0)
    if sigma2 == 0.0 or nvox < 2:
        return 0.0
    sigma, ln_n = np.sqrt(sigma2), np.log(float(nvox))
    a = np.sqrt(2. * ln_n)
    if sigma <= a:
        return float(ln_n + 0.5 * sigma2 - 7.0)
    mu = sigma * (a - (np.log(ln_n) + np.log(4. * np.pi)) / (2. * a))
    return float(mu + 6.0)


def lognormal_shift_exact(dmax, nvox):
    """The shift for a field whose maximum is known: every term is <= e^-m with m = max(0, ln nvox - LN_SUM_MAX),
    so the sum lies in [e^-m, e^LN_SUM_MAX] whatever the field looks like."""
    return float(dmax) + max(0.0, np.log(float(nvox)) - LN_SUM_MAX)


def lognormal_sums_in_range(cnt, s1, s2, esum):
    """Did the shifted exponentials of a fused log-normal P(k) stay inside the plan's floating-point range?
    Overflow shows as non-finite sums (or a sum of exponentials that is not a positive number); underflow as a
    non-empty bin whose sum of squares has lost terms, i.e. lies below (sum)^2 / n, which no set of real numbers
    does.  Bin 0 is the reference's discarded one (box.py:761-764) and is not looked at."""
    c = np.asarray(cnt)[1:]
    ok = c > 0
    a, b = np.asarray(s1)[1:][ok], np.asarray(s2)[1:][ok]
    if not (np.isfinite(esum) and esum > 0. and np.all(np.isfinite(a)) and np.all(np.isfinite(b))):
        return False
    # (a bin with exactly zero power -- a constant field: exp(d)/mean - 1 = 0, the reference returns P = 0 -- is in range: with a
    # finite, positive sum of exponentials underflow cannot produce it)
    zero = (a == 0.) & (b == 0.)
    return bool(np.all((a > 0.) | zero) and np.all(b * c[ok] >= a * a * (1. - 1e-3)))


def lognormal_sums_in_range_many(cnt, s1, s2, esum):
    """`lognormal_sums_in_range` for many records at once: s1, s2 (records, nbins), esum (records,) -> bool per record."""
    c = np.asarray(cnt)[1:]
    ok = c > 0
    a, b = np.asarray(s1)[:, 1:][:, ok], np.asarray(s2)[:, 1:][:, ok]
    with np.errstate(all="ignore"):
        fin = np.isfinite(esum) & (esum > 0.) & np.all(np.isfinite(a), axis=1) & np.all(np.isfinite(b), axis=1)
        zero = (a == 0.) & (b == 0.)
        return fin & np.all((a > 0.) | zero, axis=1) & np.all(b * c[ok] >= a * a * (1. - 1e-3), axis=1)


def lognormal_rescale(s1, s2, mean):
    """Bin sums of the transform of exp(d - shift), brought to those of exp(d)/<exp(d)> - 1: s1 / mean^2, s2 / mean^4, with
    mean = sum(exp(d - shift)) / voxels (scalar, or one per record for (records, nbins) sums).  The powers are formed by
    multiplications -- exactly rounded on every platform, for a Python float as for a numpy array, which `mean ** 4` is
    not (numpy's vectorised pow and libm's differ in the last bit on some hosts): one record finished alone and the same
    record finished in a batch give the same numbers."""
    m2 = mean * mean
    m4 = m2 * m2
    if np.ndim(mean):
        m2, m4 = m2[:, None], m4[:, None]
    return s1 / m2, s2 / m4


def finish_bins_many(cnt, s1, s2, boxfactor, eps=0.):
    """`finish_bins` for many records at once (s1, s2: (records, nbins)); the same expressions, element for element."""
    with np.errstate(all="ignore"):
        vals = s1 / (cnt * boxfactor)
        var = (s2 - s1 * s1 / cnt) / cnt
        if eps:
            var = np.where((cnt <= 2) & (var <= 4. * eps * (s1 / cnt) ** 2), 0., var)
        stddev = np.sqrt(np.maximum(var, 0.)) / boxfactor / np.sqrt(cnt)
    return vals[:, 1:], stddev[:, 1:]


def bin_edges(g, nbins=20, kbins=None):
    """Edges and the centres of bins 1..nbins-1 (box.py:745-751)."""
    if kbins is not None:
        bins = np.asarray(kbins, dtype=np.float64)
    else:
        bins = np.logspace(np.log10(g["kmin"]), np.log10(g["kmax"]), nbins)
    _b = [0.0] + list(bins)
    cent = [0.5 * (_b[j + 1] + _b[j]) for j in range(bins.size)]
    return bins, np.array(cent[1:])


def shell_thresholds(N, Lside, bins):
    """np.digitize(k, bins) as a step function of the integer shell (cubic boxes): thr[b] = first
    n^2 whose bin index exceeds b.  Shells whose |k| lies within rounding of an edge cannot be
    decided from n^2 alone (the reference's per-mode |k| differs in the last bits between
    decompositions of the same n^2) and are returned in `amb` for the exact on-device path."""
    if not np.all(np.diff(bins) >= 0):
        return None, ()
    k = shell_wavenumbers(N, Lside)
    eps = 64 * np.finfo(np.float64).eps
    lo = np.digitize(k * (1. - eps), bins)
    hi = np.digitize(k * (1. + eps), bins)
    amb = np.nonzero(lo != hi)[0]
    if amb.size > 8:
        return None, ()
    thr = np.searchsorted(hi, np.arange(1, bins.size + 1), side="left")
    return thr.astype(np.int32), tuple(int(a) for a in amb)


def finish_bins(cnt, s1, s2, boxfactor, eps=0.):
    """(mean, std/sqrt(n)) per bin from (count, sum |dk|^2, sum |dk|^4), bin 0 dropped (box.py:761-768); NaN for empty
    bins.  ``eps``: rounding unit of the |dk|^2 values the sums were formed from (2^-23 for a single-precision plan).
    A bin that holds one mode and its mirror image (count 2: the same |dk|^2 twice for a real field), or a single
    self-mirrored mode (count 1: a corner of the grid), has exactly 0 spread in the reference's np.std; the form
    sum p^2 - (sum p)^2 / n leaves +-eps p^2 of rounding there instead, whose square root would read as a spread of
    2e-4 -- so for such bins, and only there, a variance below 4 eps mean^2 is reported as 0.  Every other bin gets the variance its sums give (a spread below ~sqrt(eps) of the
    mean is beyond what sums of single-precision squares resolve; the fp64 plan resolves 1e-8)."""
    with np.errstate(all="ignore"):
        vals = s1 / (cnt * boxfactor)
        var = (s2 - s1 * s1 / cnt) / cnt
        if eps:
            var = np.where((cnt <= 2) & (var <= 4. * eps * (s1 / cnt) ** 2), 0., var)
        stddev = np.sqrt(np.maximum(var, 0.)) / boxfactor / np.sqrt(cnt)
    return np.array(vals[1:]), np.array(stddev[1:])


def bin_counts(N, Lside, bins):
    """Number of modes of the full (N,N,N) grid per np.digitize index (host, cubic boxes): used
    when no device is at hand (tests of the slab driver)."""
    m = mode_numbers(N)
    a = (m / Lside) ** 2.
    k = 2. * np.pi * np.sqrt(a[:, None, None] + a[None, :, None] + a[None, None, :])
    idx = np.digitize(k.ravel(), bins)
    return np.bincount(idx, minlength=bins.size + 1)[:bins.size].astype(np.float64)
