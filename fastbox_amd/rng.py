"""
Host model (numpy) of the device noise generator in csrc/fb_rng.h, so that a
realisation drawn on the GPU with ``rng='device'`` can be reproduced -- and
checked -- on the host: Threefry4x32-20 (Random123 constants) followed by
Box-Muller.  The fp32 device path uses hardware log2/sqrt/sin/cos, so a host
reproduction agrees to ~1e-6, the fp64 path to rounding.
"""
import numpy as np

_ROT = ((10, 26), (11, 21), (13, 27), (23, 5), (6, 20), (17, 11), (25, 10), (18, 20))
_M32 = np.uint64(0xFFFFFFFF)


def _rotl(x, r):
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & _M32


def threefry4x32_20(ctr, key):
    """ctr: 4 uint32-valued arrays (broadcastable), key: 4 ints.  Returns 4 uint64 arrays < 2^32."""
    ks = [np.uint64(int(k) & 0xFFFFFFFF) for k in key]
    ks.append(np.uint64(0x1BD11BDA) ^ ks[0] ^ ks[1] ^ ks[2] ^ ks[3])
    X = [(np.asarray(c, dtype=np.uint64) + ks[i]) & _M32 for i, c in enumerate(ctr)]
    X = list(np.broadcast_arrays(*X))
    X = [x.copy() for x in X]
    for r in range(20):
        a, b = _ROT[r % 8]
        if r % 2 == 0:
            X[0] = (X[0] + X[1]) & _M32; X[1] = _rotl(X[1], a) ^ X[0]
            X[2] = (X[2] + X[3]) & _M32; X[3] = _rotl(X[3], b) ^ X[2]
        else:
            X[0] = (X[0] + X[3]) & _M32; X[3] = _rotl(X[3], a) ^ X[0]
            X[2] = (X[2] + X[1]) & _M32; X[1] = _rotl(X[1], b) ^ X[2]
        if r % 4 == 3:
            s = (r + 1) // 4
            for i in range(4):
                X[i] = (X[i] + ks[(s + i) % 5]) & _M32
            X[3] = (X[3] + np.uint64(s)) & _M32
    return X


def box_muller(a, b, dtype=np.float64):
    dt = np.dtype(dtype).type
    u1 = (a.astype(dtype) + dt(0.5)) * dt(2.3283064365386963e-10)
    u2 = (b.astype(dtype) + dt(0.5)) * dt(2.3283064365386963e-10)
    if np.dtype(dtype) == np.float32:
        r = np.sqrt(dt(-1.3862943611198906) * np.log2(u1))
    else:
        r = np.sqrt(-2.0 * np.log(u1))
    ang = dt(2.0 * np.pi) * u2
    return (r * np.cos(ang)).astype(dtype), (r * np.sin(ang)).astype(dtype)


def half_spectrum_noise(N, seed, realisation, dtype=np.float64):
    """Complex unit-variance-per-component noise z(ix,iy,iz) for every stored mode
    (shape (N, N, N/2+1)) exactly as the device draws it; the field generator multiplies
    it by sqrt(P boxfactor) and by 1/sqrt(2) off the k_z = 0, N/2 planes."""
    nz = N // 2 + 1
    g = np.arange(N // 2, dtype=np.uint64)[:, None, None]
    iy = np.arange(N, dtype=np.uint64)[None, :, None]
    iz = np.arange(nz, dtype=np.uint64)[None, None, :]
    idx = (g * np.uint64(N) + iy) * np.uint64(nz) + iz
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, realisation & 0xFFFFFFFF, (realisation >> 32) & 0xFFFFFFFF)
    o = threefry4x32_20((idx & _M32, idx >> np.uint64(32), np.uint64(0), np.uint64(0)), key)
    a0, a1 = box_muller(o[0], o[1], dtype)
    b0, b1 = box_muller(o[2], o[3], dtype)
    z = np.empty((N, N, nz), dtype=np.complex128)
    z[:N // 2] = a0 + 1j * a1
    z[N // 2:] = b0 + 1j * b1
    return z


def los_noise(N, seed, dtype=np.float64):
    """Standard normals n(i,j,m) of the redshift-space small-scale velocities (stream 1): element
    idx of the (N,N,N) grid is output idx & 3 of call idx >> 2 (fb_rng.h los_noise_at)."""
    q = np.arange(N ** 3 // 4, dtype=np.uint64)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, 0, 0)
    o = threefry4x32_20((q & _M32, q >> np.uint64(32), np.uint64(1), np.uint64(0)), key)
    g0, g1 = box_muller(o[0], o[1], dtype)
    g2, g3 = box_muller(o[2], o[3], dtype)
    return np.stack([g0, g1, g2, g3], axis=-1).reshape(N, N, N)
