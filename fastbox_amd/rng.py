"""
Host model (numpy) of the device noise generator in csrc/fb_rng.h, so that a
realisation drawn on the GPU with ``rng='device'`` can be reproduced -- and
checked -- on the host: Philox4x32-10 (Random123 constants) followed by
Box-Muller.  The fp32 device path uses hardware log2/sqrt/sin/cos, so a host
reproduction agrees to ~1e-6, the fp64 path to rounding.

Call:  o = philox4x32_10(ctr = (idx_lo, idx_hi | stream << 24, real_lo, real_hi), key = (seed_lo, seed_hi))
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85
_S32 = np.uint64(32)


def philox4x32_10(ctr, key, rounds=10):
    """ctr: 4 uint32-valued arrays (broadcastable), key: 2 ints.  Returns 4 uint64 arrays < 2^32."""
    x = np.broadcast_arrays(*[np.asarray(c, dtype=np.uint64) for c in ctr])
    x = [a.copy() for a in x]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = _PHILOX_M0 * x[0]
        p1 = _PHILOX_M1 * x[2]
        y0 = (p1 >> _S32) ^ x[1] ^ np.uint64(k0)
        y2 = (p0 >> _S32) ^ x[3] ^ np.uint64(k1)
        x = [y0, p1 & _M32, y2, p0 & _M32]
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return x


def _call(idx, stream, seed, realisation):
    """the four words of call `idx` (uint64 array) of a stream (fb_rng.h philox_counter)"""
    idx = np.asarray(idx, dtype=np.uint64)
    c1 = (idx >> _S32) | np.uint64(int(stream) << 24)
    return philox4x32_10((idx & _M32, c1, np.uint64(realisation & 0xFFFFFFFF), np.uint64((realisation >> 32) & 0xFFFFFFFF)),
                         (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))


def box_muller(a, b, dtype=np.float64):
    dt = np.dtype(dtype).type
    u1 = (a.astype(dtype) + dt(0.5)) * dt(2.3283064365386963e-10)
    u2 = (b.astype(dtype) + dt(0.5)) * dt(2.3283064365386963e-10)
    if np.dtype(dtype) == np.float32:
        r = np.sqrt(dt(-1.3862943611198906) * np.log2(u1))
    else:
        r = np.sqrt(-2.0 * np.log(u1))
    # the device's sine / cosine take the angle in revolutions (v_sin_f32, sincospi), i.e. they see u2 itself: the
    # angle is formed in double here -- 2 pi rounded to float32 is 2.8e-8 too large, which would bias cos by that
    # much on average and, summed coherently over 10^8 modes, put a spike at the origin of a 512^3 field
    ang = 2.0 * np.pi * u2.astype(np.float64)
    return (r * np.cos(ang).astype(dtype)).astype(dtype), (r * np.sin(ang).astype(dtype)).astype(dtype)


def half_spectrum_noise(N, seed, realisation, dtype=np.float64, planes=None):
    """Complex noise z(ix,iy,iz), E|z|^2 = 1, for every stored mode (shape (N, N, N/2+1)) exactly as the
    device draws it (fb_rng.h): the field generator multiplies it by sqrt(P boxfactor).

    0 < iz < N/2: z = (g0 + i g1)/sqrt 2 of call idx = (g N + iy)(N/2+1) + iz, g = ix mod N/2, words (0,1) for
    ix < N/2, (2,3) for ix >= N/2.  The planes iz = 0, N/2 are their own mirror images and are drawn Hermitian:
    modes with iy in (0, N/2), or iy in {0, N/2} and ix in (0, N/2), as above; their mirror images
    ((N-ix)%N, (N-iy)%N) are the conjugates; the four self-mirrored modes are real, z = g0.

    ``planes``: optional iterable of iz values to draw (default: all) -- returns shape (N, N, len(planes))."""
    H = N // 2
    nz = H + 1
    izs = np.arange(nz, dtype=np.int64) if planes is None else np.asarray(list(planes), dtype=np.int64)
    ix = np.arange(N, dtype=np.int64)[:, None, None]
    iy = np.arange(N, dtype=np.int64)[None, :, None]
    iz = izs[None, None, :]
    plane = (iz == 0) | (iz == H)
    ys = (iy == 0) | (iy == H)
    xs = (ix == 0) | (ix == H)
    conj = plane & np.where(ys, ix > H, iy > H)
    real_only = plane & ys & xs
    dix = np.where(conj, (N - ix) % N, ix)
    diy = np.where(conj, (N - iy) % N, iy)
    g = dix % H
    hi = dix >= H
    idx = ((g * N + diy) * nz + iz).astype(np.uint64)
    o = _call(idx, 0, seed, realisation)
    a = np.where(hi, o[2], o[0])
    b = np.where(hi, o[3], o[1])
    g0, g1 = box_muller(a, b, dtype)
    s = np.dtype(dtype).type(np.sqrt(0.5))
    re = np.where(real_only, g0, s * g0)
    im = np.where(real_only, 0, np.where(conj, -(s * g1), s * g1))
    return re.astype(np.float64) + 1j * im.astype(np.float64)


def stream_normals(n, stream, seed, realisation=0, dtype=np.float64):
    """The first n standard normals of a single-normal stream: element idx is output idx & 3 of call idx >> 2
    (fb_rng.h stream_noise_at)."""
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    o = _call(q, stream, seed, realisation)
    g0, g1 = box_muller(o[0], o[1], dtype)
    g2, g3 = box_muller(o[2], o[3], dtype)
    return np.stack([g0, g1, g2, g3], axis=-1).reshape(-1)[:n]


def los_noise(N, seed, dtype=np.float64):
    """Standard normals n(i,j,m) of the redshift-space small-scale velocities (stream 1)."""
    return stream_normals(N ** 3, 1, seed, 0, dtype).reshape(N, N, N)
