"""
One box spread over several GPUs: slab-decomposed 3-D FFT with a single all-to-all per transform
(the reference has no distributed code -- this is the scale-out of its
realise_density / binned_power_spectrum for boxes whose Monte-Carlo needs strong scaling).

Decomposition for P ranks (one process per GPU, ``torch.distributed``; backend "nccl" = RCCL over
xGMI on the GPUs, "gloo" on CPUs for tests):

  real space : rank r owns x-planes [r N/P, (r+1) N/P)            T[N/P][N][N]
  k space    : rank r owns k_y rows [r N/P, (r+1) N/P) of every x-plane, stored x-major
               kslab = complex[N][N/P][pitch]                      (k_z fastest, Hermitian half)

  realise_density : generator fused into the x pass of the kslab  -> all-to-all (the kslab's N/P-plane
                    blocks are already contiguous: no pack)       -> y pass (reads the receive buffer
                    directly: no unpack kernel) + z c2r
  P(k)            : z r2c + y pass on the x-slab (writes the send buffer directly: no pack kernel)
                    -> all-to-all -> the receive buffer IS the kslab -> x pass with fused shell
                    binning -> all-reduce of 2*nbins+1 doubles

Each ordered pair of ranks exchanges (N/P)(N/P)(pitch) complex values; with P = 8 all 7 xGMI links of
a GPU carry one peer each.  The device noise depends on global mode indices only, so the field is
identical for every P (tests compare P = 1, 2, 4).

Communication / compute overlap, two forms.  (1) Inside ONE transform (``SlabBox(..., chunks=C)``; BASELINE config 4
is a single box): the x and y passes never mix k_z columns, so the half spectrum is cut into C chunks of k_z tile
columns, each chunk of the k-slab / exchange buffers an array of its own (equal contiguous blocks per rank), and

  compute stream :  GEN-x(0) GEN-x(1) GEN-x(2) ...  y(0) y(1) ...  z (whole rows)  y(0) y(1) ...   BIN-x(0) BIN-x(1) ...
  RCCL stream    :      a2a(0)   a2a(1)   a2a(2) ...                                  a2a(0) a2a(1) ...

the all-to-all of chunk c runs while the passes of the chunks after it (and the y passes of those before it) are
computed; only the z pass, which needs whole rows, has nothing beside it.  Fields are bit-identical for every C.
(2) Across realisations: a Monte-Carlo loop has independent realisations, so the exchanges of one are
hidden behind the passes of its neighbours -- ``realise_and_power(wait=False)`` keeps up to three realisations in
flight on this rank,

  compute stream :  GEN(i)   turn(i-1)   BIN(i-2)   GEN(i+1)   turn(i)   BIN(i-1)  ...
  RCCL stream    :     x1(i) ........ x2(i-1) .........   x1(i+1) ...... x2(i) ....

(GEN = generator + x pass of the k-slab, x1 = its all-to-all, turn = y pass, fused z passes, y pass of the x-slab,
x2 = the all-to-all back, BIN = x pass with the shell binning): both all-to-alls of a realisation run as asynchronous
collectives while the compute stream works on another realisation, with three rotating buffer pairs.  The kernels are
those of the synchronous path, the fields are bit-identical, and the z pass is covered as well.
Bin sums stay on the device (a ring of records, all-reduced asynchronously) until ``result()`` is called.

The per-rank arithmetic is behind a small "ops" interface: ``HipSlabOps`` drives libfastbox_hip (no
CPU fallback); tests inject a numpy implementation to exercise the exchange logic under gloo.
"""
import ctypes

import numpy as np

from . import hostgeom


class HipSlabOps(object):
    """Per-rank HIP operations on torch CUDA tensors (data_ptr() -> C ABI, torch's current stream)."""

    def __init__(self, geom, nparts, part, precision="f32", device=0):
        import torch
        from . import _lib
        from .device import Engine
        self.torch, self._lib = torch, _lib
        self.g, self.P, self.part = geom, nparts, part
        self.N = geom["N"]
        self.dev = torch.device("cuda", device)
        axis2, ksc, kpar = hostgeom.axis_tables(self.N, geom["L"])
        self.engine = Engine(self.N, geom["L"], axis2, ksc, kpar, geom["z"], precision=precision, device=device)
        self.rdtype = torch.float32 if precision == "f32" else torch.float64
        self.pitch, self.rows = self.engine.pitch, self.engine.rows
        self.nloc = self.N // nparts

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def _call(self, name, *args):
        self._lib.call(name, self.engine._plan, *args)

    # buffers (complex stored as trailing dimension of 2 reals)
    def new_real(self):
        return self.torch.empty((self.nloc, self.N, self.N), dtype=self.rdtype, device=self.dev)

    def new_half_local(self):
        return self.torch.empty((self.nloc, self.rows, self.pitch, 2), dtype=self.rdtype, device=self.dev)

    def new_kslab(self):
        return self.torch.empty((self.N, self.nloc, self.pitch, 2), dtype=self.rdtype, device=self.dev)

    def new_results(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64, device=self.dev)

    # tables
    def set_amplitude(self, amp_shells):
        self.engine.set_amplitude_shells(amp_shells)

    def set_bins(self, bins, thr, amb):
        self.engine.set_bins(bins, thr, amb)

    def set_exp_shift(self, shift):
        self.engine._set_exp_shift(shift)

    def bin_counts(self):
        return self.engine.bin_counts()

    def max_real(self, real):
        """Largest value of this rank's real slab (a device reduction of torch's; the caller all-reduces it)."""
        return float(real.max().item())

    # transforms
    def x_generate(self, kslab, seed, realisation):
        self._call("fb_slab_x_generate", kslab.data_ptr(), self.P, self.part, seed & (2 ** 64 - 1),
                   realisation & (2 ** 64 - 1), self._stream())

    def unpack(self, recv, half_local):
        self._call("fb_slab_unpack", recv.data_ptr(), half_local.data_ptr(), self.P, self._stream())

    def inverse_local(self, half_local, real):
        self._call("fb_slab_inverse_local", half_local.data_ptr(), real.data_ptr(), self.P, self._stream())

    def forward_local(self, real, half_local, pre_exp, expsum):
        self._call("fb_slab_forward_local", real.data_ptr(), half_local.data_ptr(), self.P, 1 if pre_exp else 0,
                   expsum.data_ptr() if pre_exp else None, self._stream())

    fused_exchange = True          # the y pass writes / reads the all-to-all buffer itself (no pack / unpack kernels)

    def forward_packed(self, real, half_local, send, pre_exp, expsum):
        self._call("fb_slab_forward_packed", real.data_ptr(), half_local.data_ptr(), send.data_ptr(), self.P,
                   1 if pre_exp else 0, expsum.data_ptr() if pre_exp else None, self._stream())

    def inverse_packed(self, recv, half_local, real):
        self._call("fb_slab_inverse_packed", recv.data_ptr(), half_local.data_ptr(), real.data_ptr(), self.P,
                   self._stream())

    def turnaround(self, recv, half_local, real, send, pre_exp, expsum):
        self._call("fb_slab_turnaround", recv.data_ptr(), half_local.data_ptr(), real.data_ptr(), send.data_ptr(),
                   self.P, 1 if pre_exp else 0, expsum.data_ptr() if pre_exp else None, self._stream())

    def pack(self, half_local, send):
        self._call("fb_slab_pack", half_local.data_ptr(), send.data_ptr(), self.P, self._stream())

    # the same passes over a range of k_z tile columns (a chunk lives in an array of its own, see fastbox_hip.h)
    def tile_geometry(self):
        """(k_z columns per tile of the strided passes, tiles per row of the half spectrum)"""
        tz, nt = ctypes.c_int(), ctypes.c_int()
        self._call("fb_slab_tile_geometry", ctypes.byref(tz), ctypes.byref(nt))
        return tz.value, nt.value

    def new_chunk_store(self, total_columns):
        """flat buffer for the chunk arrays of one exchange side: [N][N/P][total_columns] complex in all"""
        return self.torch.empty((self.N * self.nloc * total_columns * 2,), dtype=self.rdtype, device=self.dev)

    def x_generate_chunk(self, kchunk, seed, realisation, tile0, ntile):
        self._call("fb_slab_x_generate_chunk", kchunk.data_ptr(), self.P, self.part, seed & (2 ** 64 - 1),
                   realisation & (2 ** 64 - 1), tile0, ntile, self._stream())

    def y_inverse_chunk(self, recv_chunk, half_local, tile0, ntile):
        self._call("fb_slab_y_inverse_chunk", recv_chunk.data_ptr(), half_local.data_ptr(), self.P, tile0, ntile,
                   self._stream())

    def y_forward_chunk(self, half_local, send_chunk, tile0, ntile):
        self._call("fb_slab_y_forward_chunk", half_local.data_ptr(), send_chunk.data_ptr(), self.P, tile0, ntile,
                   self._stream())

    def z_pass(self, half_local, real, which, pre_exp=False, expsum=None):
        """which: 0 half -> real, 1 real -> half (of exp(real) if pre_exp), 2 half -> real -> half of (exp of) it"""
        self._call("fb_slab_z_pass", half_local.data_ptr(), real.data_ptr(), self.P, which, 1 if pre_exp else 0,
                   expsum.data_ptr() if (pre_exp and expsum is not None) else None, self._stream())

    def x_bin_chunk(self, kchunk, tile0, ntile, first, last, results):
        self._call("fb_slab_x_bin_chunk", kchunk.data_ptr(), self.P, self.part, tile0, ntile, 1 if first else 0,
                   1 if last else 0, results.data_ptr(), self._stream())

    def x_bin(self, kslab, results):
        self._call("fb_slab_x_bin", kslab.data_ptr(), self.P, self.part, results.data_ptr(), self._stream())


class SlabBox(object):
    """The slab-decomposed counterpart of ``CosmoBox.realise_density`` / ``binned_power_spectrum``
    for cubic boxes with the device RNG.  Collective: every rank of the group calls each method."""

    def __init__(self, cosmo, box_scale=1e3, nsamp=512, redshift=0., precision="f32", seed=0,
                 rank=None, world=None, group=None, ops_factory=None, device=None, pk_fn=None, chunks=1, comm=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        # comm: an object with  all_to_all(rank, send, recv) -> handle with wait()  and  all_reduce(rank, tensor)  that stands
        # in for torch.distributed (VirtualComm below: several ranks of one process, so that the sequencing of this class --
        # asynchronous chunk exchanges, waits, buffer reuse -- is what a single-GPU test runs)
        self._comm = comm
        if rank is None:
            rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank, self.world = rank, world
        self._eps = 2. ** -23 if (precision == "f32" and ops_factory is None) else 2. ** -52     # hostgeom.finish_bins
        self.g = hostgeom.grid(box_scale, nsamp)
        N = self.N = nsamp
        if not self.g["cubic"]:
            raise ValueError("SlabBox needs a cubic box")
        if N % world:
            raise ValueError("the number of ranks must divide nsamp")
        self.boxfactor, self.kmin, self.kmax = self.g["boxfactor"], self.g["kmin"], self.g["kmax"]
        self.seed, self._realisation = int(seed), 0
        if pk_fn is None:
            from . import box as _box
            if isinstance(cosmo, dict):
                cosmo = _box._ccl.Cosmology(**cosmo)
            a = 1. / (1. + redshift)
            pk_fn = lambda k: _box._ccl.nonlin_matter_power(cosmo, k=k, a=a)
        self.cosmo = cosmo
        if ops_factory is None:
            ops_factory = lambda g, P, r: HipSlabOps(g, P, r, precision=precision,
                                                     device=(r if device is None else device))
        self.ops = ops_factory(self.g, world, rank)
        if isinstance(comm, str):
            if comm != "rccl":
                raise ValueError("comm: None (torch.distributed), 'rccl' (the library's own communicator) or an object")
            # the data path's collectives inside libfastbox_hip.so; the id travels by fastbox_amd.rendezvous (a one-rank box
            # makes a real one-rank RCCL communicator, so that this path can be exercised on a single GPU)
            self._comm = RcclComm.from_environment(self.ops.engine, world, rank, self.ops._stream) if world > 1 else \
                RcclComm(self.ops.engine, 1, 0, RcclComm.unique_id(), self.ops._stream)
        amp = hostgeom.shell_amplitude(N, self.g["L"][0], self.boxfactor, pk_fn)
        self.ops.set_amplitude(amp)
        # fused log-normal transforms form exp(d - shift): same estimate, and with the shift taken from the variance of
        # the fields this box draws the sums stay in range on single-precision plans (hostgeom.lognormal_shift; a
        # realisation whose extremes fall outside is repeated with the shift from its own maximum, _redo_lognormal)
        self._sigma2 = hostgeom.field_variance_cubic(N, amp)
        self._ln_shift = hostgeom.lognormal_shift(self._sigma2, float(N) ** 3)
        self.ln_repeats = 0
        if hasattr(self.ops, "set_exp_shift"):
            self.ops.set_exp_shift(self._ln_shift)
        self._kslab = self.ops.new_kslab()
        self._xbuf = self.ops.new_kslab()          # same byte count as [P][N/P][N/P][pitch]
        self._half = self.ops.new_half_local()
        self.delta_x = None
        self._bin_cache, self._bins_set = {}, None
        # pipelined Monte-Carlo steps (realise_and_power(wait=False)): rotating buffer pairs, the real slab and the
        # ring of result records are allocated once, on first use
        self._pairs = None
        self._real_mc = None
        self._ring, self._ring_next = None, 0
        self._inflight, self._retired = [], []     # _Ticket objects, oldest first
        self._submitted = 0                        # tickets handed out so far: ticket i owns buffer pair i % 3
        # ONE transform in `chunks` pieces along k_z (the x and y passes never mix k_z columns): the all-to-all of a
        # chunk runs while the passes of the next are computed -- what hides the exchange of a single box (BASELINE
        # config 4), where there is no neighbouring realisation to hide it behind
        self.chunks = 1
        self._chunk_tab = None
        if hasattr(self.ops, "tile_geometry"):
            tz, ntile = self.ops.tile_geometry()
            C = max(1, min(int(chunks or 1), ntile))
            sizes = [ntile // C + (1 if i < ntile % C else 0) for i in range(C)]
            starts = [sum(sizes[:i]) for i in range(C)]
            self.chunks, self._chunk_tab, self._chunk_tz = C, list(zip(starts, sizes)), tz
            self._chunk_cols = ntile * tz
            self._cstore = [None, None]            # the two exchange sides, as chunk arrays (allocated on first use)

    def _fused(self, call):
        """Run the exchange-buffer-addressing form of a y pass if the backend has it and the rank count allows it
        (it must divide the points a thread holds of a line); False = use pack / unpack instead."""
        if not getattr(self.ops, "fused_exchange", False):
            return False
        try:
            call()
            return True
        except Exception as e:                      # FastBoxError(FB_ERR_UNSUPPORTED): remember and fall back
            if getattr(e, "code", None) != -3:
                raise
            self.ops.fused_exchange = False
            return False

    # -- the single data-path collective --------------------------------------------------
    def _host_staged(self, t):
        """gloo has no device all-to-all: stage through the host (tests on a single GPU only)."""
        return self.world > 1 and t.is_cuda and self._dist.get_backend(self.group) == "gloo"

    def _exchange(self, send, recv):
        """All-to-all of equal blocks; returns the buffer that holds the result (`send` itself for one rank)."""
        if self.world == 1:
            return send
        if self._comm is not None:
            self._comm.all_to_all(self.rank, send, recv).wait()
            return recv
        if self._host_staged(send):
            h_in = send.detach().cpu().view(-1)
            h_out = h_in.new_empty(h_in.shape)
            self._dist.all_to_all_single(h_out, h_in, group=self.group)
            recv.view(-1).copy_(h_out)
        else:
            self._dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group)
        return recv

    def _all_reduce(self, t):
        if self.world == 1:
            return
        if self._comm is not None:
            self._comm.all_reduce(self.rank, t)
            return
        if self._host_staged(t):
            h = t.detach().cpu()
            self._dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            self._dist.all_reduce(t, group=self.group)

    # -- realise_density (box.py:130-194, throughput mode) ------------------------------------
    def _gen_local(self):
        self.ops.x_generate(self._kslab, self.seed, self._realisation)
        self._realisation += 1
        return self._kslab

    def _gen_finish(self, recv):
        real = self.ops.new_real()
        if not self._fused(lambda: self.ops.inverse_packed(recv, self._half, real)):
            self.ops.unpack(recv, self._half)
            self.ops.inverse_local(self._half, real)
        self.delta_x = real
        return real

    # -- one transform in k_z chunks (self.chunks > 1) ---------------------------------------------
    def _chunk_views(self, side):
        """The chunk arrays [N][N/P][W_c] (x 2 reals) of exchange side 0 / 1: views into one flat allocation."""
        if self._cstore[side] is None:
            flat = self.ops.new_chunk_store(self._chunk_cols)
            views, off = [], 0
            for t0, nt in self._chunk_tab:
                n = self.N * (self.N // self.world) * nt * self._chunk_tz * 2
                views.append(flat[off:off + n])
                off += n
            self._cstore[side] = (flat, views)
        return self._cstore[side][1]

    def _exchange_chunk(self, send, recv):
        """Start the all-to-all of one chunk array; returns (handle or None, buffer that will hold the result)."""
        if self.world == 1:
            return None, send
        return self._exchange_async(send, recv), recv

    def _inverse_chunked(self):
        """generator + x pass, exchange, y pass -- chunk by chunk; leaves the x-slab's half spectrum in self._half"""
        A, B = self._chunk_views(0), self._chunk_views(1)
        moves = []
        for c, (t0, nt) in enumerate(self._chunk_tab):
            self.ops.x_generate_chunk(A[c], self.seed, self._realisation, t0, nt)
            moves.append(self._exchange_chunk(A[c], B[c]))          # runs beside the generator pass of chunk c + 1
        self._realisation += 1
        for c, (t0, nt) in enumerate(self._chunk_tab):
            w, buf = moves[c]
            if w is not None:
                w.wait()
            self.ops.y_inverse_chunk(buf, self._half, t0, nt)       # ... and beside the y passes of the chunks before

    def _forward_chunked(self, nb):
        """y pass, exchange, x pass with the binning -- chunk by chunk, from self._half; returns the bin sums"""
        A, B = self._chunk_views(0), self._chunk_views(1)
        moves = []
        for c, (t0, nt) in enumerate(self._chunk_tab):
            self.ops.y_forward_chunk(self._half, A[c], t0, nt)
            moves.append(self._exchange_chunk(A[c], B[c]))
        last = len(self._chunk_tab) - 1
        for c, (t0, nt) in enumerate(self._chunk_tab):
            w, buf = moves[c]
            if w is not None:
                w.wait()
            self.ops.x_bin_chunk(buf, t0, nt, c == 0, c == last, self._res)
        return self._res

    def realise_density(self):
        """This rank's x-slab of delta_x (kept in ``self.delta_x``)."""
        self._drain()          # tickets in flight read / write the buffers this call is about to use
        if self.chunks > 1:
            self._inverse_chunked()
            real = self.ops.new_real()
            self.ops.z_pass(self._half, real, 0)
            self.delta_x = real
            return real
        send = self._gen_local()
        return self._gen_finish(self._exchange(send, self._xbuf))

    # -- binned_power_spectrum (box.py:696-768) ----------------------------------------------
    def _pk_setup(self, nbins, kbins):
        key = ("n", int(nbins)) if kbins is None else ("k", np.asarray(kbins, dtype=np.float64).tobytes())
        hit = self._bin_cache.get(key)
        if hit is None:                       # np.digitize over every shell: once per bin set, not per realisation
            bins, kc = hostgeom.bin_edges(self.g, nbins, kbins)
            thr, amb = hostgeom.shell_thresholds(self.N, self.g["L"][0], bins)
            if thr is None:
                raise ValueError("these bin edges cannot be expressed as shell thresholds")
            if len(self._bin_cache) > 16:
                self._bin_cache.clear()
            hit = self._bin_cache[key] = (bins, kc, thr, amb)
        bins, kc, thr, amb = hit
        if self._bins_set != key:
            self.ops.set_bins(bins, thr, amb)
            self._bins_set = key
        return bins, kc.copy()

    def _pk_local(self, real, lognormal, nb):
        self._res = self.ops.new_results(2 * nb + 1)
        if not self._fused(lambda: self.ops.forward_packed(real, self._half, self._xbuf, lognormal,
                                                             self._res[2 * nb:])):
            self.ops.forward_local(real, self._half, lognormal, self._res[2 * nb:])
            self.ops.pack(self._half, self._xbuf)
        return self._xbuf

    def _pk_finish(self, kslab, nb):
        self.ops.x_bin(kslab, self._res)
        return self._res

    def realise_and_power(self, nbins=20, kbins=None, lognormal=False, wait=True):
        """``realise_density()`` followed by ``binned_power_spectrum(lognormal=...)`` of the new field, as the
        Monte-Carlo loop does, with the z passes of the two fused: the real slab is written once, not read back.
        Returns (kc, pk, stddev); ``self.delta_x`` holds the slab.

        ``wait=False``: returns a ticket at once; its ``result()`` gives the triple.  With several ranks up to three
        realisations are in flight and their all-to-alls overlap the passes of the others (module docstring);
        ``flush()`` issues whatever is still outstanding, ``result()`` does so implicitly.  In this mode
        ``self.delta_x`` is ONE buffer shared by all tickets: it holds the slab of whichever realisation went through
        its z pass last, and the next one overwrites it (copy it after ``flush()`` to keep a field).  Any other
        method of the box first issues what is in flight (``_drain``), so calls may be interleaved freely."""
        if not wait and self.world > 1:
            return self._submit(nbins, kbins, lognormal)
        self._drain()
        bins, kc = self._pk_setup(nbins, kbins)
        nb = bins.size
        if lognormal and not bins[0] > 0.:
            raise ValueError("the fused log-normal P(k) needs kbins[0] > 0")
        if self.chunks > 1:
            self._inverse_chunked()
            self._res = self.ops.new_results(2 * nb + 1)
            real = self.ops.new_real()
            self.ops.z_pass(self._half, real, 2, lognormal, self._res[2 * nb:])     # delta_x written once, not read back
            self.delta_x = real
            res = self._forward_chunked(nb)
            idx = self._realisation - 1
            redo = lambda: self._redo_lognormal(None, nbins, kbins, realisation=idx)
            if not wait:
                return _Deferred(self, res, kc, nb, lognormal, redo)
            return self._finish_power(res, kc, nb, lognormal, redo)
        recv = self._exchange(self._gen_local(), self._xbuf)
        self._res = self.ops.new_results(2 * nb + 1)
        real = self.ops.new_real()
        send = self._kslab if recv is self._xbuf else self._xbuf           # the buffer the exchange did not return
        if self._fused(lambda: self.ops.turnaround(recv, self._half, real, send, lognormal, self._res[2 * nb:])):
            self.delta_x = real
        else:
            real = self._gen_finish(recv)
            send = self._pk_local(real, lognormal, nb)
        other = self._kslab if send is self._xbuf else self._xbuf
        res = self._pk_finish(self._exchange(send, other), nb)
        idx = self._realisation - 1
        redo = lambda: self._redo_lognormal(None, nbins, kbins, realisation=idx)
        if not wait:
            return _Deferred(self, res, kc, nb, lognormal, redo)
        return self._finish_power(res, kc, nb, lognormal, redo)

    # -- pipelined Monte-Carlo steps ----------------------------------------------------------
    RING = 64                                      # result records kept on the device before they must be fetched

    def _exchange_async(self, send, recv):
        """All-to-all as an asynchronous collective: returns a handle whose ``wait()`` makes the compute stream wait
        for it (None: nothing to wait for).  gloo with device tensors (single-GPU rehearsal) stages synchronously."""
        if self._comm is not None:
            return self._comm.all_to_all(self.rank, send, recv)
        if self._host_staged(send):
            self._exchange(send, recv)
            return None
        return self._dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group, async_op=True)

    def _all_reduce_async(self, t):
        if self._comm is not None or self._host_staged(t):
            self._all_reduce(t)
            return None
        return self._dist.all_reduce(t, group=self.group, async_op=True)

    def _mc_buffers(self, nb):
        if self._pairs is None:
            # pair 0 reuses the synchronous path's buffers; a realisation owns its pair from GEN to BIN
            self._pairs = [(self._kslab, self._xbuf)] + [(self.ops.new_kslab(), self.ops.new_kslab()) for _ in range(2)]
            self._real_mc = self.ops.new_real()
        if self._ring is None or self._ring.shape[1] != 2 * nb + 1:
            self._drain()
            self._fetch_retired()
            self._ring = self.ops.new_results(self.RING * (2 * nb + 1)).view(self.RING, 2 * nb + 1)
            self._ring_next = 0

    def _turn(self, t):
        if t.w1 is not None:
            t.w1.wait()
        a, b = t.pair
        real = self._real_mc
        if not self._fused(lambda: self.ops.turnaround(b, self._half, real, a, t.lognormal, t.res[2 * t.nb:])):
            self.ops.unpack(b, self._half)
            self.ops.inverse_local(self._half, real)
            self.ops.forward_local(real, self._half, t.lognormal, t.res[2 * t.nb:])
            self.ops.pack(self._half, a)
        self.delta_x = real
        t.w2 = self._exchange_async(a, b)
        t.state = "turned"

    def _bin(self, t):
        if t.w2 is not None:
            t.w2.wait()
        self.ops.x_bin(t.pair[1], t.res)
        t.w3 = self._all_reduce_async(t.res)
        t.state = "binned"

    def _drain(self):
        """Issue the remaining stages of every realisation in flight, oldest first."""
        for t in self._inflight:
            if t.state == "gen":
                self._turn(t)
        for t in self._inflight:
            if t.state == "turned":
                self._bin(t)
        self._retired += self._inflight
        self._inflight = []

    def _fetch_retired(self):
        for t in self._retired:
            t.fetch()
        self._retired = []

    def _submit(self, nbins, kbins, lognormal):
        bins, kc = self._pk_setup(nbins, kbins)
        nb = bins.size
        if lognormal and not bins[0] > 0.:
            raise ValueError("the fused log-normal P(k) needs kbins[0] > 0")
        self._mc_buffers(nb)
        if self._ring_next and self._ring_next % self.RING == 0:
            self._drain()                                  # the ring is about to wrap: fetch what is pending first
            self._fetch_retired()
        # (the pair follows the submission count, not the realisation index: a caller may set `_realisation` to any
        # value -- a resumed Monte-Carlo run does -- and two tickets in flight must never share a pair)
        tk = _Ticket(self, kc, nb, lognormal, self._pairs[self._submitted % 3],
                     self._ring[self._ring_next % self.RING], self._realisation, (nbins, kbins))
        self._submitted += 1
        self._ring_next += 1
        # compute stream: GEN(i), turn(i-1), BIN(i-2); the collectives follow their producers on the RCCL stream
        self.ops.x_generate(tk.pair[0], self.seed, self._realisation)
        self._realisation += 1
        tk.w1 = self._exchange_async(tk.pair[0], tk.pair[1])
        fl = self._inflight
        if fl and fl[-1].state == "gen":
            self._turn(fl[-1])
        if len(fl) > 1 and fl[-2].state == "turned":
            self._bin(fl[-2])
        fl.append(tk)
        while fl and fl[0].state == "binned":              # its buffer pair is free again
            self._retired.append(fl.pop(0))
        return tk

    def flush(self):
        """Issue the remaining stages of every realisation submitted with ``wait=False``."""
        self._drain()

    def _finish_power(self, res, kc, nb, lognormal, redo=None):
        self._all_reduce(res)                                     # 2*nbins+1 doubles
        return self._finish_host(res.detach().cpu().numpy(), kc, nb, lognormal, redo)

    def _finish_host(self, h, kc, nb, lognormal, redo=None):
        """(kc, pk, stddev) from the all-reduced record; a log-normal record whose exponentials left the plan's
        floating-point range (hostgeom.lognormal_sums_in_range) is formed again by `redo` with the exact shift."""
        cnt = self.ops.bin_counts()
        for attempt in (0, 1):
            s1, s2, esum = h[0:2 * nb:2].copy(), h[1:2 * nb:2].copy(), h[2 * nb]
            if not lognormal or hostgeom.lognormal_sums_in_range(cnt, s1, s2, esum):
                break
            if attempt or redo is None:
                raise FloatingPointError("log-normal P(k): the exponentials left the plan's floating-point range "
                                         "(sum = %r)" % (esum,))
            h = redo()
        if lognormal:
            mean = esum / float(self.N) ** 3
            s1, s2 = hostgeom.lognormal_rescale(s1, s2, mean)
        return (kc,) + hostgeom.finish_bins(cnt, s1, s2, self.boxfactor, self._eps)

    def binned_power_spectrum(self, delta_x=None, nbins=20, kbins=None, lognormal=False):
        """P(k) of the distributed field (of its log-normal transform if ``lognormal``); every rank
        returns the full (kc, pk, stddev) triple."""
        self._drain()          # tickets in flight read / write the buffers this call is about to use
        real = self.delta_x if delta_x is None else delta_x
        bins, kc = self._pk_setup(nbins, kbins)
        nb = bins.size
        if lognormal and not bins[0] > 0.:
            raise ValueError("the fused log-normal P(k) needs kbins[0] > 0")
        res = self._pk_of(real, lognormal, nb)
        return self._finish_power(res, kc, nb, lognormal, redo=lambda: self._redo_lognormal(real, nbins, kbins))

    def _pk_of(self, real, lognormal, nb):
        """bin sums (not yet all-reduced) of the distributed field `real` (of exp of it if `lognormal`)"""
        if self.chunks > 1:
            self._res = self.ops.new_results(2 * nb + 1)
            self.ops.z_pass(self._half, real, 1, lognormal, self._res[2 * nb:])
            return self._forward_chunked(nb)
        send = self._pk_local(real, lognormal, nb)
        return self._pk_finish(self._exchange(send, self._kslab), nb)

    def _redo_lognormal(self, real, nbins, kbins, realisation=None):
        """Bin sums of the log-normal transform of `real` (or of realisation number `realisation`, drawn again) with
        the shift taken from the field's own maximum: the repeat of a step whose exponentials left the plan's
        floating-point range.  Collective (the sums that decide it are all-reduced, so every rank gets here)."""
        self._drain()
        if realisation is not None:
            keep = self._realisation
            self._realisation = realisation
            real = self.realise_density()
            self._realisation = keep
        import torch
        m = torch.tensor([self.ops.max_real(real)], dtype=torch.float64)
        if self.world > 1 and self._comm is not None:
            m = m.to(real.device)
            self._comm.all_reduce(self.rank, m, op="max")
        elif self.world > 1:
            if self._dist.get_backend(self.group) != "gloo":
                m = m.to(real.device)
            self._dist.all_reduce(m, op=self._dist.ReduceOp.MAX, group=self.group)
        self.ops.set_exp_shift(hostgeom.lognormal_shift_exact(float(m.item()), float(self.N) ** 3))
        try:
            bins, kc = self._pk_setup(nbins, kbins)
            nb = bins.size
            res = self._pk_of(real, True, nb)
            self._all_reduce(res)
            h = res.detach().cpu().numpy()
        finally:
            self.ops.set_exp_shift(self._ln_shift)
        self.ln_repeats += 1
        return h


class _Ticket(object):
    """One pipelined realisation + P(k) of a SlabBox (see SlabBox._submit)."""

    def __init__(self, box, kc, nb, lognormal, pair, res, realisation, bin_args):
        self.box, self.kc, self.nb, self.lognormal, self.pair, self.res = box, kc, nb, lognormal, pair, res
        self.realisation, self.bin_args = realisation, bin_args
        self.state, self.w1, self.w2, self.w3 = "gen", None, None, None
        self._host = None

    def fetch(self):
        if self._host is None:
            if self.state != "binned":
                self.box._drain()
            if self.w3 is not None:
                self.w3.wait()
            self._host = self.res.detach().cpu().numpy().copy()
            if self in self.box._retired:
                self.box._retired.remove(self)
        return self._host

    def result(self):
        box = self.box
        redo = lambda: box._redo_lognormal(None, self.bin_args[0], self.bin_args[1], realisation=self.realisation)
        return box._finish_host(self.fetch(), self.kc, self.nb, self.lognormal, redo)


class _Deferred(object):
    """wait=False on a single rank: the step has been queued, the bin sums are fetched on ``result()``."""

    def __init__(self, box, res, kc, nb, lognormal, redo=None):
        self.box, self.res, self.kc, self.nb, self.lognormal, self.redo = box, res, kc, nb, lognormal, redo

    def result(self):
        return self.box._finish_power(self.res, self.kc, self.nb, self.lognormal, self.redo)


class RcclComm(object):
    """The collectives of a SlabBox through the library's own communicator (C ABI: fb_comm_create, fb_slab_exchange_begin /
    _wait, fb_allreduce_f64 -- RCCL over xGMI on a stream the library owns, event hand-off with the compute stream): ctypes
    only, no torch.distributed.  `stream_fn()` gives the stream the rank's passes run on; buffers are anything with
    data_ptr() / numel() / element_size() (torch tensors here)."""

    class _Handle(object):
        def __init__(self, comm, ticket):
            self.comm, self.ticket = comm, ticket

        def wait(self):
            c = self.comm
            c._lib.call("fb_slab_exchange_wait", c._plan, self.ticket, c._stream_fn())

    def __init__(self, engine, world, rank, unique_id=None, stream_fn=None):
        from . import _lib
        self._lib, self._plan, self.world, self.rank = _lib, engine._plan, int(world), int(rank)
        self._stream_fn = stream_fn or (lambda: engine.stream)
        self._engine = engine                     # keeps the plan alive
        if unique_id is not None and len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        buf = ctypes.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
        _lib.call("fb_comm_create", self._plan, self.world, self.rank, buf)

    @staticmethod
    def unique_id():
        from . import _lib
        buf = ctypes.create_string_buffer(128)
        _lib.call("fb_comm_unique_id", buf)
        return buf.raw

    @classmethod
    def from_environment(cls, engine, world, rank, stream_fn=None):
        """Rank 0 draws the id, fastbox_amd.rendezvous hands it to the others (MASTER_ADDR / MASTER_PORT of the launcher)."""
        from .rendezvous import broadcast_bytes
        uid = broadcast_bytes(cls.unique_id() if rank == 0 else None, rank, world)
        return cls(engine, world, rank, uid, stream_fn)

    def info(self):
        w, r, v = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        self._lib.call("fb_comm_info", self._plan, ctypes.byref(w), ctypes.byref(r), ctypes.byref(v))
        return {"world": w.value, "rank": r.value, "rccl_version": v.value}

    def all_to_all(self, rank, send, recv):
        nbytes = send.numel() * send.element_size()
        if nbytes % self.world or recv.numel() * recv.element_size() != nbytes:
            raise ValueError("all-to-all of equal blocks: buffers of the same size, divisible by the number of ranks")
        t = ctypes.c_int32()
        self._lib.call("fb_slab_exchange_begin", self._plan, send.data_ptr(), recv.data_ptr(), nbytes // self.world,
                       self._stream_fn(), ctypes.byref(t))
        return RcclComm._Handle(self, t.value)

    def all_reduce(self, rank, t, op="sum"):
        if t.element_size() != 8 or not t.is_floating_point():
            raise TypeError("fb_allreduce_f64 reduces float64 device arrays")
        self._lib.call("fb_allreduce_f64", self._plan, t.data_ptr(), t.numel(), 0 if op == "sum" else 1, self._stream_fn())

    def close(self):
        if self._plan is not None:
            self._lib.call("fb_comm_destroy", self._plan)
            self._plan = None


class VirtualComm(object):
    """The collectives of P SlabBox ranks that live in ONE process, each driven by its own thread (``run``): an all-to-all
    returns a handle at once, like an asynchronous collective; ``wait()`` of a rank's k-th collective blocks until every
    rank has posted its k-th, copies that rank's blocks out of the posted send buffers, and blocks again until every rank
    has copied (a send buffer may be reused after its own wait, as with a real collective).  All tensor work is enqueued on
    the device's current stream, so stream order = the order the threads were let through."""

    class _Handle(object):
        def __init__(self, comm, rank, k, recv):
            self.comm, self.rank, self.k, self.recv = comm, rank, k, recv

        def wait(self):
            c = self.comm
            c._meet()                                     # every rank has posted (and enqueued what produces) collective k
            sends = c._posted[self.k]
            flat = self.recv.view(c.P, -1)
            for q in range(c.P):
                flat[q].copy_(sends[q].view(c.P, -1)[self.rank])
            c._meet()                                     # every rank has taken its blocks: the send buffers are free again

    def __init__(self, P):
        import threading
        self.P = P
        self._bar = threading.Barrier(P)
        self._posted = {}
        self._count = [0] * P
        self._lock = threading.Lock()
        self._red = {}

    def _meet(self):
        self._bar.wait(timeout=600)

    def all_to_all(self, rank, send, recv):
        k = self._count[rank]
        self._count[rank] += 1
        with self._lock:
            self._posted.setdefault(k, [None] * self.P)[rank] = send
        return VirtualComm._Handle(self, rank, k, recv)

    def all_reduce(self, rank, t, op="sum"):
        with self._lock:
            self._red.setdefault("cur", [None] * self.P)[rank] = t
        self._meet()
        parts = self._red["cur"]
        total = parts[0].clone()
        for q in range(1, self.P):                         # the same order on every rank
            total = total + parts[q] if op == "sum" else total.maximum(parts[q])
        self._meet()
        t.copy_(total)
        self._meet()
        if rank == 0:
            self._red.pop("cur", None)
        self._meet()

    def run(self, boxes, fn):
        """fn(box) on every box, one thread per rank; returns the results in rank order (re-raises the first error)."""
        import threading
        out, err = [None] * self.P, []

        def work(r):
            try:
                out[r] = fn(boxes[r])
            except BaseException as e:                   # noqa: B902 -- a failed rank must release the others
                err.append(e)
                self._bar.abort()
        ts = [threading.Thread(target=work, args=(r,)) for r in range(self.P)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if err:
            raise err[0]
        return out


def run_virtual(boxes, fn_local, fn_finish):
    """Drive several SlabBox objects that live in ONE process (virtual ranks, e.g. to exercise the
    slab kernels on a single GPU): phase 1 on every rank, the all-to-all by block copies, phase 2."""
    P = len(boxes)
    sends = [fn_local(b) for b in boxes]
    recvs = []
    for r, b in enumerate(boxes):                  # the whole exchange first: phase 2 may reuse the send buffers
        recv = b._xbuf if sends[r] is not b._xbuf else b._kslab
        flat = recv.view(P, -1)
        for q in range(P):
            flat[q].copy_(sends[q].view(P, -1)[r])
        recvs.append(recv)
    return [fn_finish(b, recv) for b, recv in zip(boxes, recvs)]


def run_virtual_chunked(boxes, nb, lognormal):
    """The chunked ``realise_and_power`` of several SlabBox objects that live in ONE process (virtual ranks on a single
    GPU): every chunk's all-to-all is done by block copies.  Leaves each box's x-slab in ``delta_x`` and returns the
    per-rank bin-sum records (to be added up, as the all-reduce would)."""
    P = len(boxes)
    tab = boxes[0]._chunk_tab
    A = [b._chunk_views(0) for b in boxes]
    B = [b._chunk_views(1) for b in boxes]

    def exchange(c):
        for r in range(P):
            dst = B[r][c].view(P, -1)
            for q in range(P):
                dst[q].copy_(A[q][c].view(P, -1)[r])
    for b, a in zip(boxes, A):
        for c, (t0, nt) in enumerate(tab):
            b.ops.x_generate_chunk(a[c], b.seed, b._realisation, t0, nt)
        b._realisation += 1
    for c in range(len(tab)):
        exchange(c)
    for b, bb in zip(boxes, B):
        for c, (t0, nt) in enumerate(tab):
            b.ops.y_inverse_chunk(bb[c], b._half, t0, nt)
        b._res = b.ops.new_results(2 * nb + 1)
        b.delta_x = b.ops.new_real()
        b.ops.z_pass(b._half, b.delta_x, 2, lognormal, b._res[2 * nb:])
        for c, (t0, nt) in enumerate(tab):
            b.ops.y_forward_chunk(b._half, A[boxes.index(b)][c], t0, nt)
    for c in range(len(tab)):
        exchange(c)
    last = len(tab) - 1
    for b, bb in zip(boxes, B):
        for c, (t0, nt) in enumerate(tab):
            b.ops.x_bin_chunk(bb[c], t0, nt, c == 0, c == last, b._res)
    return [b._res for b in boxes]
