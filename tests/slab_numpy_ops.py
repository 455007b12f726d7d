"""Test double for the per-rank operations of fastbox_amd.distributed.SlabBox: the same interface
as HipSlabOps, implemented with numpy on CPU torch tensors, so that the slab exchange logic can
be exercised under gloo without a GPU.  (Test infrastructure; never used by the product.)"""
import numpy as np
import torch

from fastbox_amd import hostgeom, rng


class NumpySlabOps(object):
    def __init__(self, geom, nparts, part):
        self.g, self.P, self.part = geom, nparts, part
        self.N = N = geom["N"]
        self.nz = N // 2 + 1
        self.pitch = (self.nz + 15) // 16 * 16
        self.rows = N + 1
        self.nloc = N // nparts

    def new_real(self):
        return torch.zeros((self.nloc, self.N, self.N), dtype=torch.float64)

    def new_half_local(self):
        return torch.zeros((self.nloc, self.rows, self.pitch, 2), dtype=torch.float64)

    def new_kslab(self):
        return torch.zeros((self.N, self.nloc, self.pitch, 2), dtype=torch.float64)

    def new_results(self, n):
        return torch.zeros(n, dtype=torch.float64)

    @staticmethod
    def _c(t):
        return t.numpy().view(np.complex128)[..., 0]

    def set_amplitude(self, amp):
        self.amp = np.asarray(amp)

    def set_bins(self, bins, thr, amb):
        self.bins = np.asarray(bins)

    def set_exp_shift(self, shift):
        self.exp_shift = float(shift)

    def max_real(self, real):
        return float(real.numpy().max())

    def bin_counts(self):
        return hostgeom.bin_counts(self.N, self.g["L"][0], self.bins)

    def _n2(self):
        m = hostgeom.mode_numbers(self.N).astype(np.int64)
        ky = m[self.part * self.nloc:(self.part + 1) * self.nloc]
        return (m[:, None, None] ** 2 + ky[None, :, None] ** 2 + m[None, None, :self.nz] ** 2)

    def x_generate(self, kslab, seed, realisation):
        N, nz = self.N, self.nz
        z = rng.half_spectrum_noise(N, seed, realisation)[:, self.part * self.nloc:(self.part + 1) * self.nloc, :]
        H = z * self.amp[self._n2()]
        self._c(kslab)[:, :, :nz] = np.fft.ifft(H, axis=0)

    def unpack(self, recv, half_local):
        r = self._c(recv).reshape(self.P, self.nloc, self.nloc, self.pitch)
        h = self._c(half_local)
        for q in range(self.P):
            h[:, q * self.nloc:(q + 1) * self.nloc, :] = r[q]

    def inverse_local(self, half_local, real):
        h = self._c(half_local)[:, :self.N, :self.nz]
        real.numpy()[...] = np.fft.irfft(np.fft.ifft(h, axis=1), n=self.N, axis=2)

    def forward_local(self, real, half_local, pre_exp, expsum):
        f = real.numpy()
        if pre_exp:
            f = np.exp(f - getattr(self, "exp_shift", 0.0))
            expsum.numpy()[0] = f.sum()
        self._c(half_local)[:, :self.N, :self.nz] = np.fft.fft(np.fft.rfft(f, axis=2), axis=1)

    def pack(self, half_local, send):
        s = self._c(send).reshape(self.P, self.nloc, self.nloc, self.pitch)
        h = self._c(half_local)
        for q in range(self.P):
            s[q] = h[:, q * self.nloc:(q + 1) * self.nloc, :]

    def x_bin(self, kslab, results):
        nb = self.bins.size
        X = np.fft.fft(self._c(kslab)[:, :, :self.nz], axis=0)
        p = np.abs(X) ** 2
        k = 2. * np.pi * np.sqrt(self._n2().astype(np.float64)) / self.g["L"][0]
        idx = np.digitize(k.ravel(), self.bins)
        w = np.full(self.nz, 2.0); w[0] = w[-1] = 1.0
        wp = (p * w[None, None, :]).ravel()
        wp2 = (p * p * w[None, None, :]).ravel()
        out = results.numpy()
        out[0:2 * nb:2] = np.bincount(idx, weights=wp, minlength=nb + 1)[:nb]
        out[1:2 * nb:2] = np.bincount(idx, weights=wp2, minlength=nb + 1)[:nb]

    # ---- the chunked interface (a range of k_z tile columns at a time; a chunk is an array of its own) ----
    TZ = 2                       # columns per tile of this test double (the HIP kernels use 16 / 8 / 4 / 2)

    def tile_geometry(self):
        return self.TZ, (self.nz + self.TZ - 1) // self.TZ

    def new_chunk_store(self, total_columns):
        return torch.zeros((self.N * self.nloc * total_columns * 2,), dtype=torch.float64)

    def _chunk(self, t, lead, tile0, ntile):
        """complex view [lead...][W] of a flat chunk array, its first absolute column, and how many of its columns exist"""
        W = ntile * self.TZ
        a = t.numpy().view(np.complex128).reshape(lead + (W,))
        col0 = tile0 * self.TZ
        return a, col0, max(0, min(W, self.nz - col0))

    def x_generate_chunk(self, kchunk, seed, realisation, tile0, ntile):
        a, col0, nv = self._chunk(kchunk, (self.N, self.nloc), tile0, ntile)
        z = rng.half_spectrum_noise(self.N, seed, realisation)[:, self.part * self.nloc:(self.part + 1) * self.nloc, :]
        H = z * self.amp[self._n2()]
        a[:, :, :nv] = np.fft.ifft(H, axis=0)[:, :, col0:col0 + nv]

    def y_inverse_chunk(self, recv_chunk, half_local, tile0, ntile):
        # (layout only: this double takes the y transform together with the z transform, in z_pass)
        r, col0, nv = self._chunk(recv_chunk, (self.P, self.nloc, self.nloc), tile0, ntile)
        h = self._c(half_local)
        for q in range(self.P):
            h[:, q * self.nloc:(q + 1) * self.nloc, col0:col0 + nv] = r[q][:, :, :nv]

    def y_forward_chunk(self, half_local, send_chunk, tile0, ntile):
        sd, col0, nv = self._chunk(send_chunk, (self.P, self.nloc, self.nloc), tile0, ntile)
        h = self._c(half_local)
        for q in range(self.P):
            sd[q][:, :, :nv] = h[:, q * self.nloc:(q + 1) * self.nloc, col0:col0 + nv]

    def z_pass(self, half_local, real, which, pre_exp=False, expsum=None):
        if which in (0, 2):
            self.inverse_local(half_local, real)
        if which in (1, 2):
            self.forward_local(real, half_local, pre_exp, expsum)

    def x_bin_chunk(self, kchunk, tile0, ntile, first, last, results):
        a, col0, nv = self._chunk(kchunk, (self.N, self.nloc), tile0, ntile)
        nb = self.bins.size
        X = np.fft.fft(a[:, :, :nv], axis=0)
        p = np.abs(X) ** 2
        k = 2. * np.pi * np.sqrt(self._n2()[:, :, col0:col0 + nv].astype(np.float64)) / self.g["L"][0]
        idx = np.digitize(k.ravel(), self.bins)
        w = np.full(self.nz, 2.0); w[0] = w[-1] = 1.0
        w = w[col0:col0 + nv]
        out = results.numpy()
        if first:
            out[0:2 * nb] = 0.0
        out[0:2 * nb:2] += np.bincount(idx, weights=(p * w[None, None, :]).ravel(), minlength=nb + 1)[:nb]
        out[1:2 * nb:2] += np.bincount(idx, weights=(p * p * w[None, None, :]).ravel(), minlength=nb + 1)[:nb]
