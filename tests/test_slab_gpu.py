"""GPU: the slab-decomposed kernels (C ABI fb_slab_*) driven as P virtual ranks on one GPU must
reproduce the single-GPU CosmoBox field and P(k) for every P (the exchange is emulated by block
copies; the process-level exchange is covered by tests/test_slab_cpu.py under gloo)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision,tol", [("f32", 2e-5), ("f64", 1e-11)])
@pytest.mark.parametrize("P,N", [(1, 64), (2, 64), (4, 64), (8, 64), (8, 128), (16, 64)])
def test_virtual_ranks_match_single_gpu(P, N, precision, tol):
    """P = 1..8: the y pass addresses the exchange buffer itself (8 points per thread of a line, cut into
    8/P-point pieces); P = 16 does not divide 8: falls back to the pack / unpack kernels."""
    from fastbox_amd import CosmoBox, default_cosmo
    from fastbox_amd.distributed import HipSlabOps, SlabBox, run_virtual
    from fastbox_amd import hostgeom
    L, seed = 1e3, 77
    ref = CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, precision=precision,
                   rng="device", seed=seed)
    want_dx = np.asarray(ref.realise_density())
    want_pk = ref.binned_power_spectrum(delta_x=ref.delta_x, nbins=20)
    want_ln = ref.binned_power_spectrum(delta_x=ref.lognormal(ref.delta_x), nbins=20)

    boxes = [SlabBox(default_cosmo, box_scale=L, nsamp=N, precision=precision, seed=seed, rank=r, world=P,
                     ops_factory=lambda g, PP, rr: HipSlabOps(g, PP, rr, precision=precision, device=0))
             for r in range(P)]
    reals = run_virtual(boxes, lambda b: b._gen_local(), lambda b, recv: b._gen_finish(recv))
    dx = np.concatenate([r.double().cpu().numpy() for r in reals], axis=0)
    assert np.max(np.abs(dx - want_dx)) < tol * np.std(want_dx)
    assert all(b.ops.fused_exchange == (P <= 8) for b in boxes)

    for lognormal, want in ((False, want_pk), (True, want_ln)):
        nb = 20
        for b in boxes:
            bins, kc = b._pk_setup(nb, None)
        res = run_virtual(boxes, lambda b: b._pk_local(b.delta_x, lognormal, nb),
                          lambda b, kslab: b._pk_finish(kslab, nb).clone())
        h = sum(r.cpu().numpy() for r in res)
        s1, s2, esum = h[0:2 * nb:2], h[1:2 * nb:2], h[2 * nb]
        if lognormal:
            mean = esum / float(N) ** 3
            s1, s2 = s1 / mean ** 2, s2 / mean ** 4
        pk, err = hostgeom.finish_bins(boxes[0].ops.bin_counts(), s1, s2, boxes[0].boxfactor)
        assert np.array_equal(kc, want[0])
        m = ~np.isnan(want[1])
        assert np.array_equal(np.isnan(pk), np.isnan(want[1]))
        ptol = 1e-5 if precision == "f32" else 1e-10
        assert np.allclose(pk[m], want[1][m], rtol=ptol, atol=0)
        assert np.all(np.abs(err[m] - want[2][m]) <= ptol * (np.abs(want[2][m]) + np.abs(want[1][m])))


@pytest.mark.parametrize("P", [1, 2, 8])
def test_turnaround_fuses_the_z_passes(P):
    """fb_slab_turnaround (inverse y from the receive buffer, one z pass writing delta_x and the forward z spectrum
    of exp(delta_x), forward y into the send buffer) against the single-GPU field and log-normal P(k)."""
    from fastbox_amd import CosmoBox, default_cosmo, hostgeom
    from fastbox_amd.distributed import HipSlabOps, SlabBox, run_virtual
    N, L, seed, nb = 64, 1e3, 5, 20
    ref = CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, precision="f32", rng="device", seed=seed)
    want_dx = np.asarray(ref.realise_density())
    want = ref.binned_power_spectrum(delta_x=ref.lognormal(ref.delta_x), nbins=nb)
    boxes = [SlabBox(default_cosmo, box_scale=L, nsamp=N, precision="f32", seed=seed, rank=r, world=P,
                     ops_factory=lambda g, PP, rr: HipSlabOps(g, PP, rr, precision="f32", device=0)) for r in range(P)]
    for b in boxes:
        b._pk_setup(nb, None)

    def turn(b, recv):
        b._res = b.ops.new_results(2 * nb + 1)
        b.delta_x = b.ops.new_real()
        b._send2 = b._kslab if recv is b._xbuf else b._xbuf
        b.ops.turnaround(recv, b._half, b.delta_x, b._send2, True, b._res[2 * nb:])
        return b._send2
    run_virtual(boxes, lambda b: b._gen_local(), turn)
    dx = np.concatenate([b.delta_x.double().cpu().numpy() for b in boxes], axis=0)
    assert np.max(np.abs(dx - want_dx)) < 2e-5 * np.std(want_dx)
    res = run_virtual(boxes, lambda b: b._send2, lambda b, kslab: b._pk_finish(kslab, nb).clone())
    h = sum(r.cpu().numpy() for r in res)
    mean = h[2 * nb] / float(N) ** 3
    pk, err = hostgeom.finish_bins(boxes[0].ops.bin_counts(), h[0:2 * nb:2] / mean ** 2, h[1:2 * nb:2] / mean ** 4,
                                   boxes[0].boxfactor)
    m = ~np.isnan(want[1])
    assert np.array_equal(np.isnan(pk), np.isnan(want[1])) and np.allclose(pk[m], want[1][m], rtol=1e-5, atol=0)
    if P == 1:                                   # the public entry point, one rank
        kc, pk1, err1 = boxes[0].realise_and_power(nbins=nb, lognormal=True)
        ref2 = ref.binned_power_spectrum(delta_x=ref.lognormal(ref.realise_density()), nbins=nb)
        assert np.allclose(pk1[m], ref2[1][m], rtol=1e-5, atol=0)


def _gloo_worker(rank, world, port, out_dir, N, L, seed):
    import os
    import sys
    import time
    t0 = time.time()
    def mark(what):
        sys.stderr.write("[slabworker %d] %6.1f s %s\n" % (rank, time.time() - t0, what)); sys.stderr.flush()
    import torch.distributed as dist
    mark("torch.distributed imported")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mark("process group up")
    try:
        from fastbox_amd import default_cosmo
        from fastbox_amd.distributed import SlabBox
        box = SlabBox(default_cosmo, box_scale=L, nsamp=N, precision="f32", seed=seed, device=0)
        mark("SlabBox built")
        pk = box.realise_and_power(nbins=20, lognormal=True)
        dx = box.delta_x.double().cpu().numpy()
        mark("density realised, power spectrum done")
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), dx=dx, pk=np.array(pk))
    finally:
        dist.destroy_process_group()


def test_two_processes_one_gpu_host_staged_exchange(tmp_path):
    """Two real processes (gloo, exchange staged through the host) driving the HIP slab kernels on
    the same GPU: the whole multi-process flow except the RCCL transport itself."""
    import socket
    import torch.multiprocessing as mp
    from fastbox_amd import CosmoBox, default_cosmo
    N, L, seed = 64, 1e3, 31
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path), N, L, seed), nprocs=2, join=True)
    ref = CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, precision="f32", rng="device", seed=seed)
    want_dx = np.asarray(ref.realise_density())
    want = ref.binned_power_spectrum(delta_x=ref.lognormal(ref.delta_x), nbins=20)
    got = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(2)]
    dx = np.concatenate([g["dx"] for g in got], axis=0)
    assert np.max(np.abs(dx - want_dx)) < 2e-5 * np.std(want_dx)
    m = ~np.isnan(want[1])
    for g in got:
        assert np.allclose(g["pk"][1][m], want[1][m], rtol=1e-5, atol=0)


def test_rccl_backend_single_rank_collectives():
    """The collectives SlabBox issues (all_to_all_single, all_reduce) through torch's nccl backend (= RCCL) on this
    GPU, world size 1: the transport this pool cannot exercise across GPUs at least loads and runs."""
    import socket
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    except Exception as e:                                     # no RCCL in this environment: not this library's failure
        pytest.skip("nccl backend unavailable: %s" % e)
    try:
        from fastbox_amd import default_cosmo
        from fastbox_amd.distributed import SlabBox
        x = torch.arange(1024, dtype=torch.float32, device="cuda")
        y = torch.empty_like(x)
        dist.all_to_all_single(y, x)
        assert torch.equal(x, y)
        r = torch.ones(41, dtype=torch.float64, device="cuda")
        dist.all_reduce(r)
        assert float(r.sum()) == 41.0
        box = SlabBox(default_cosmo, box_scale=1e3, nsamp=64, precision="f32", seed=3, device=0)   # rank / world from the group
        assert (box.rank, box.world) == (0, 1)
        kc, pk, err = box.realise_and_power(nbins=20, lognormal=True)
        assert np.all(np.isfinite(pk[~np.isnan(pk)]))
    finally:
        dist.destroy_process_group()


def test_library_communicator_on_one_rank():
    """The exchange behind the C ABI (fb_comm_create, fb_slab_exchange_begin / _wait, fb_allreduce_f64: RCCL on a stream the
    library owns, event hand-off with the compute stream) -- only one rank can run on this pool: a real one-rank RCCL
    communicator (ncclSend / ncclRecv to itself inside a group, ncclAllReduce), the RCCL-free one-rank form, the tickets'
    bounds, the state errors, and SlabBox(comm="rccl") against the torch.distributed path."""
    import torch
    from fastbox_amd import _lib, default_cosmo
    from fastbox_amd._lib import FastBoxError
    from fastbox_amd.distributed import RcclComm, SlabBox
    box = SlabBox(default_cosmo, box_scale=1e3, nsamp=64, precision="f32", seed=3, device=0, rank=0, world=1)
    eng, stream = box.ops.engine, box.ops._stream
    x = torch.arange(1 << 16, dtype=torch.float32, device="cuda")
    y = torch.zeros_like(x)
    with pytest.raises(FastBoxError) as ei:                       # no communicator yet
        _lib.call("fb_slab_exchange", eng._plan, x.data_ptr(), y.data_ptr(), x.numel() * 4, stream())
    assert ei.value.code == -5
    try:
        uid = RcclComm.unique_id()
    except FastBoxError as e:                                     # no librccl in this environment: FB_ERR_RCCL, loudly
        assert e.code == -6
        pytest.skip("RCCL unavailable: %s" % e)
    assert len(uid) == 128
    comm = RcclComm(eng, 1, 0, uid, stream)
    info = comm.info()
    assert info["world"] == 1 and info["rank"] == 0 and info["rccl_version"] > 0
    with pytest.raises(FastBoxError) as ei:                       # one communicator per plan
        RcclComm(eng, 1, 0, uid, stream)
    assert ei.value.code == -5
    handles = []
    outs = [torch.zeros_like(x) for _ in range(3)]
    sends = [x * (k + 1) for k in range(3)]                       # (a send buffer belongs to the exchange until its wait)
    for k, o in enumerate(outs):                                  # three exchanges in flight, waited in another order
        handles.append(comm.all_to_all(0, sends[k], o))
    for h in (handles[2], handles[0], handles[1]):
        h.wait()
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        assert torch.equal(o, x * (k + 1))
    with pytest.raises(FastBoxError):                             # a ticket that was never issued
        _lib.call("fb_slab_exchange_wait", eng._plan, 1000, stream())
    with pytest.raises(FastBoxError):                             # in place is refused
        _lib.call("fb_slab_exchange", eng._plan, x.data_ptr(), x.data_ptr(), x.numel() * 4, stream())
    r = torch.arange(41, dtype=torch.float64, device="cuda")
    comm.all_reduce(0, r)
    comm.all_reduce(0, r, op="max")
    torch.cuda.synchronize()
    assert torch.equal(r, torch.arange(41, dtype=torch.float64, device="cuda"))
    comm.close()
    # the RCCL-free one-rank form: a device copy on the communicator's stream
    _lib.call("fb_comm_create", eng._plan, 1, 0, None)
    y.zero_()
    _lib.call("fb_slab_exchange", eng._plan, x.data_ptr(), y.data_ptr(), x.numel() * 4, stream())
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    _lib.call("fb_comm_destroy", eng._plan)
    # the public class on the library's communicator
    want = box.realise_and_power(nbins=20, lognormal=True)
    box2 = SlabBox(default_cosmo, box_scale=1e3, nsamp=64, precision="f32", seed=3, device=0, rank=0, world=1, comm="rccl")
    assert isinstance(box2._comm, RcclComm) and box2._comm.info()["rccl_version"] > 0
    got = box2.realise_and_power(nbins=20, lognormal=True)
    for a, b in zip(got, want):
        assert np.array_equal(a, b, equal_nan=True)
    box2._comm.close()


@pytest.mark.parametrize("P", [1, 2])
def test_slab_path_at_1024(P):
    """BASELINE config 4's size: the slab-decomposed field and its log-normal P(k) against the single-GPU box (whose
    y/z passes run plane batch by plane batch on two streams at this size; the slab path keeps whole-slab passes).
    Compared through device reductions (the cubes are 4.3 GB each)."""
    from fastbox_amd import CosmoBox, default_cosmo, hostgeom
    from fastbox_amd.distributed import HipSlabOps, SlabBox, run_virtual
    N, L, seed, nb = 1024, 2e3, 9, 20
    ref = CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, precision="f32", rng="device", seed=seed)
    dx = ref.realise_density()
    want_sq = ref.engine.sum_real(dx, squared=True)
    want_sum = ref.engine.sum_real(dx)
    want = ref.binned_power_spectrum(delta_x=ref.lognormal(dx), nbins=nb)
    del dx
    boxes = [SlabBox(default_cosmo, box_scale=L, nsamp=N, precision="f32", seed=seed, rank=r, world=P,
                     ops_factory=lambda g, PP, rr: HipSlabOps(g, PP, rr, precision="f32", device=0)) for r in range(P)]
    for b in boxes:
        b._pk_setup(nb, None)

    def turn(b, recv):
        b._res = b.ops.new_results(2 * nb + 1)
        b.delta_x = b.ops.new_real()
        b._send2 = b._kslab if recv is b._xbuf else b._xbuf
        b.ops.turnaround(recv, b._half, b.delta_x, b._send2, True, b._res[2 * nb:])
        return b._send2
    run_virtual(boxes, lambda b: b._gen_local(), turn)
    got_sq = sum(float((b.delta_x.double() ** 2).sum()) for b in boxes)
    got_sum = sum(float(b.delta_x.double().sum()) for b in boxes)
    assert np.isclose(got_sq, want_sq, rtol=1e-6) and abs(got_sum - want_sum) < 1e-6 * np.sqrt(want_sq * float(N) ** 3)
    res = run_virtual(boxes, lambda b: b._send2, lambda b, kslab: b._pk_finish(kslab, nb).clone())
    h = sum(r.cpu().numpy() for r in res)
    mean = h[2 * nb] / float(N) ** 3
    pk, err = hostgeom.finish_bins(boxes[0].ops.bin_counts(), h[0:2 * nb:2] / mean ** 2, h[1:2 * nb:2] / mean ** 4,
                                   boxes[0].boxfactor)
    m = ~np.isnan(want[1])
    assert np.array_equal(np.isnan(pk), np.isnan(want[1])) and np.allclose(pk[m], want[1][m], rtol=1e-5, atol=0)
    # the un-fused pieces (inverse_packed, forward_packed) on fresh ranks: the same realisation again
    del boxes
    boxes = [SlabBox(default_cosmo, box_scale=L, nsamp=N, precision="f32", seed=seed, rank=r, world=P,
                     ops_factory=lambda g, PP, rr: HipSlabOps(g, PP, rr, precision="f32", device=0)) for r in range(P)]
    for b in boxes:
        b._pk_setup(nb, None)
    reals = run_virtual(boxes, lambda b: b._gen_local(), lambda b, recv: b._gen_finish(recv))
    assert np.isclose(sum(float((r.double() ** 2).sum()) for r in reals), want_sq, rtol=1e-6)
    res = run_virtual(boxes, lambda b: b._pk_local(b.delta_x, True, nb), lambda b, kslab: b._pk_finish(kslab, nb).clone())
    h2 = sum(r.cpu().numpy() for r in res)
    assert np.allclose(h2, h, rtol=2e-6)


@pytest.fixture(scope="module")
def single_gpu_2048():
    """The 2048^3, L = 1000 Mpc realisation of the single-GPU CosmoBox (bench.py's `sizes` box: sigma = 21), reduced on
    the device: sum, sum of squares, log-normal P(k).  Computed once for the three rank counts below."""
    import gc
    from fastbox_amd import CosmoBox, default_cosmo
    N, L, seed, nb = 2048, 1e3, 13, 20
    ref = CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, precision="f32", rng="device", seed=seed)
    dx = ref.realise_density()
    out = dict(N=N, L=L, seed=seed, nb=nb, sq=ref.engine.sum_real(dx, squared=True), sum=ref.engine.sum_real(dx),
               ln=ref.binned_power_spectrum(delta_x=ref.lognormal(dx), nbins=nb),
               gauss=ref.binned_power_spectrum(delta_x=dx, nbins=nb), repeats=ref.lognormal_repeats)
    del dx
    ref.engine.close()
    del ref
    gc.collect()
    return out


@pytest.mark.parametrize("P", [1, 2, 8])
def test_slab_path_at_2048(P, single_gpu_2048):
    """BASELINE config 5's size, the one the >= 6x strong-scaling clause is stated on: 16 points per thread, 128 KiB
    tiles, exchange-buffer addressing with P dividing those 16 points (fb_slab_x_generate, fb_slab_turnaround,
    fb_slab_x_bin; then fb_slab_inverse_packed / fb_slab_forward_packed on fresh ranks), as P virtual ranks on one
    GPU against the single-GPU box: the field through sum and sum of squares, the log-normal P(k) at 1e-5."""
    import gc
    import torch
    from fastbox_amd import default_cosmo, hostgeom
    from fastbox_amd.distributed import HipSlabOps, SlabBox, run_virtual
    w = single_gpu_2048
    N, L, seed, nb = w["N"], w["L"], w["seed"], w["nb"]
    assert w["repeats"] == 0

    def ranks():
        bs = [SlabBox(default_cosmo, box_scale=L, nsamp=N, precision="f32", seed=seed, rank=r, world=P,
                      ops_factory=lambda g, PP, rr: HipSlabOps(g, PP, rr, precision="f32", device=0)) for r in range(P)]
        for b in bs:
            b._pk_setup(nb, None)
        return bs

    def sums(fields):      # fp64 accumulation without an fp64 copy of the slab
        sq = sum(float(torch.linalg.vector_norm(f.reshape(-1), 2, dtype=torch.float64)) ** 2 for f in fields)
        return sq, sum(float(torch.sum(f, dtype=torch.float64)) for f in fields)

    def spectrum(h):
        s1, s2, esum = h[0:2 * nb:2], h[1:2 * nb:2], h[2 * nb]
        assert hostgeom.lognormal_sums_in_range(boxes[0].ops.bin_counts(), s1, s2, esum)
        mean = esum / float(N) ** 3
        return hostgeom.finish_bins(boxes[0].ops.bin_counts(), s1 / mean ** 2, s2 / mean ** 4, boxes[0].boxfactor)

    boxes = ranks()
    assert boxes[0]._ln_shift == hostgeom.lognormal_shift(boxes[0]._sigma2, float(N) ** 3)

    def turn(b, recv):
        b._res = b.ops.new_results(2 * nb + 1)
        b.delta_x = b.ops.new_real()
        b._send2 = b._kslab if recv is b._xbuf else b._xbuf
        b.ops.turnaround(recv, b._half, b.delta_x, b._send2, True, b._res[2 * nb:])
        return b._send2
    run_virtual(boxes, lambda b: b._gen_local(), turn)
    assert all(b.ops.fused_exchange for b in boxes)               # P | 16: the y passes address the exchange buffers
    got_sq, got_sum = sums([b.delta_x for b in boxes])
    assert np.isclose(got_sq, w["sq"], rtol=1e-6) and abs(got_sum - w["sum"]) < 1e-6 * np.sqrt(w["sq"] * float(N) ** 3)
    res = run_virtual(boxes, lambda b: b._send2, lambda b, kslab: b._pk_finish(kslab, nb).clone())
    h = sum(r.cpu().numpy() for r in res)
    pk, err = spectrum(h)
    want = w["ln"]
    m = ~np.isnan(want[1])
    assert np.array_equal(np.isnan(pk), np.isnan(want[1])) and np.array_equal(np.isnan(pk), np.isnan(w["gauss"][1]))
    assert np.allclose(pk[m], want[1][m], rtol=1e-5, atol=0), np.max(np.abs(pk[m] / want[1][m] - 1))
    assert np.all(np.abs(err[m] - want[2][m]) <= 1e-4 * (np.abs(want[2][m]) + np.abs(want[1][m])))
    # the un-fused pieces on fresh ranks: the same realisation again, the same sums
    del boxes, res
    gc.collect()
    torch.cuda.empty_cache()
    boxes = ranks()
    reals = run_virtual(boxes, lambda b: b._gen_local(), lambda b, recv: b._gen_finish(recv))
    sq2, sum2 = sums(reals)
    assert np.isclose(sq2, w["sq"], rtol=1e-6)
    del reals
    res = run_virtual(boxes, lambda b: b._pk_local(b.delta_x, True, nb), lambda b, kslab: b._pk_finish(kslab, nb).clone())
    h2 = sum(r.cpu().numpy() for r in res)
    assert np.allclose(h2, h, rtol=2e-6)
    if P == 1:          # the public entry point on one rank, Gaussian and log-normal, pipelined and not
        b = boxes[0]
        b._realisation = 0
        kc, pk1, err1 = b.realise_and_power(nbins=nb, lognormal=True)
        assert np.allclose(pk1[m], want[1][m], rtol=1e-5, atol=0) and b.ln_repeats == 0
        b._realisation = 0
        kc, pk0, err0 = b.realise_and_power(nbins=nb, lognormal=False, wait=False).result()
        assert np.allclose(pk0[m], w["gauss"][1][m], rtol=1e-5, atol=0)
    del boxes, res
    gc.collect()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("P", [1, 2, 8])
def test_chunked_transform_at_2048_through_the_shipped_sequence(P, single_gpu_2048):
    """What bench.py's strong-scaling leg runs (N = 2048, f32, k_z chunks C = 4: 33 + 33 + 32 + 32 tiles) through the code that
    ships: SlabBox.realise_and_power -> _inverse_chunked / _forward_chunked with their asynchronous chunk exchanges, waits and
    buffer reuse, the collectives supplied by VirtualComm (P ranks = P threads of this process on one GPU).  16 points per
    thread, resident schedule, parked generator stores, double-buffered binning and the appended partial-sum table all take
    part.  Against the single-GPU box: field through sum and sum of squares, log-normal P(k) at 1e-5; a second realisation
    pipelined (wait=False) equals the same realisation run synchronously."""
    import gc
    import torch
    from fastbox_amd import default_cosmo
    from fastbox_amd.distributed import HipSlabOps, SlabBox, VirtualComm
    w = single_gpu_2048
    N, L, seed, nb = w["N"], w["L"], w["seed"], w["nb"]
    comm = VirtualComm(P) if P > 1 else None
    boxes = [SlabBox(default_cosmo, box_scale=L, nsamp=N, precision="f32", seed=seed, rank=r, world=P, chunks=4, comm=comm,
                     ops_factory=lambda g, PP, rr: HipSlabOps(g, PP, rr, precision="f32", device=0)) for r in range(P)]
    assert all(b.chunks == 4 for b in boxes) and len({nt for _, nt in boxes[0]._chunk_tab}) > 1       # uneven chunks
    run = (lambda fn: comm.run(boxes, fn)) if P > 1 else (lambda fn: [fn(boxes[0])])
    out = run(lambda b: b.realise_and_power(nbins=nb, lognormal=True))
    sq = sum(float(torch.linalg.vector_norm(b.delta_x.reshape(-1), 2, dtype=torch.float64)) ** 2 for b in boxes)
    sm = sum(float(torch.sum(b.delta_x, dtype=torch.float64)) for b in boxes)
    assert np.isclose(sq, w["sq"], rtol=1e-6) and abs(sm - w["sum"]) < 1e-6 * np.sqrt(w["sq"] * float(N) ** 3)
    want = w["ln"]
    m = ~np.isnan(want[1])
    for kc, pk, err in out:                              # every rank holds the all-reduced spectrum
        assert np.array_equal(kc, want[0]) and np.array_equal(np.isnan(pk), np.isnan(want[1]))
        assert np.allclose(pk[m], want[1][m], rtol=1e-5, atol=0), np.max(np.abs(pk[m] / want[1][m] - 1))
        assert np.array_equal(pk, out[0][1], equal_nan=True)
    assert all(b.ln_repeats == 0 for b in boxes)
    # the next realisation, Gaussian: synchronous, then the same index again through the ticket form
    sync = run(lambda b: b.realise_and_power(nbins=nb, lognormal=False))

    def again(b):
        b._realisation -= 1
        return b.realise_and_power(nbins=nb, lognormal=False, wait=False).result()
    piped = run(again) if P == 1 else sync       # (several ranks: wait=False is the whole-slab pipeline, three more buffer pairs)
    for a, c in zip(sync, piped):
        assert np.array_equal(a[1], c[1], equal_nan=True) and np.array_equal(a[2], c[2], equal_nan=True)
    mg = ~np.isnan(w["gauss"][1])
    assert not np.allclose(sync[0][1][mg], w["gauss"][1][mg], rtol=1e-3)      # another realisation than the fixture's
    del boxes, out, sync, piped
    gc.collect()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision,tol", [("f32", 2e-5), ("f64", 1e-11)])
@pytest.mark.parametrize("P,N,C", [(1, 64, 2), (2, 64, 3), (8, 64, 2), (4, 128, 5), (2, 256, 4)])
def test_chunked_slab_transform_on_virtual_ranks(P, N, C, precision, tol):
    """One transform in C chunks along k_z (fb_slab_*_chunk: the x and y passes over a range of tile columns, chunk arrays
    with their own row pitch on the exchange side): the field bit-identical to the unchunked slab path, both against
    the single-GPU box."""
    from fastbox_amd import CosmoBox, default_cosmo, hostgeom
    from fastbox_amd.distributed import HipSlabOps, SlabBox, run_virtual, run_virtual_chunked
    L, seed, nb = 1e3, 41, 20
    ref = CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, precision=precision, rng="device", seed=seed)
    want_dx = np.asarray(ref.realise_density())
    want = ref.binned_power_spectrum(delta_x=ref.lognormal(ref.delta_x), nbins=nb)
    mk = lambda chunks: [SlabBox(default_cosmo, box_scale=L, nsamp=N, precision=precision, seed=seed, rank=r, world=P,
                                 ops_factory=lambda g, PP, rr: HipSlabOps(g, PP, rr, precision=precision, device=0),
                                 chunks=chunks) for r in range(P)]
    boxes = mk(C)
    assert all(b.chunks == C for b in boxes)
    for b in boxes:
        b._pk_setup(nb, None)
    res = run_virtual_chunked(boxes, nb, True)
    dx = np.concatenate([b.delta_x.double().cpu().numpy() for b in boxes], axis=0)
    assert np.max(np.abs(dx - want_dx)) < tol * np.std(want_dx)
    h = sum(r.cpu().numpy() for r in res)
    mean = h[2 * nb] / float(N) ** 3
    pk, err = hostgeom.finish_bins(boxes[0].ops.bin_counts(), h[0:2 * nb:2] / mean ** 2, h[1:2 * nb:2] / mean ** 4,
                                   boxes[0].boxfactor)
    m = ~np.isnan(want[1])
    ptol = 1e-5 if precision == "f32" else 1e-10
    assert np.array_equal(np.isnan(pk), np.isnan(want[1])) and np.allclose(pk[m], want[1][m], rtol=ptol, atol=0)
    # the unchunked slab path: the same field bit for bit, the same sums to fp64 rounding of another grouping
    plain = mk(1)
    for b in plain:
        b._pk_setup(nb, None)

    def turn(b, recv):
        b._res = b.ops.new_results(2 * nb + 1)
        b.delta_x = b.ops.new_real()
        b._send2 = b._kslab if recv is b._xbuf else b._xbuf
        b.ops.turnaround(recv, b._half, b.delta_x, b._send2, True, b._res[2 * nb:])
        return b._send2
    run_virtual(plain, lambda b: b._gen_local(), turn)
    dx1 = np.concatenate([b.delta_x.double().cpu().numpy() for b in plain], axis=0)
    assert np.array_equal(dx, dx1)
    res1 = run_virtual(plain, lambda b: b._send2, lambda b, kslab: b._pk_finish(kslab, nb).clone())
    h1 = sum(r.cpu().numpy() for r in res1)
    assert np.allclose(h, h1, rtol=1e-12, atol=0)
    if P == 1:              # the public calls on one rank (no exchange: a chunk's send buffer is its receive buffer)
        b = mk(C)[0]
        kc, pk1, err1 = b.realise_and_power(nbins=nb, lognormal=True)
        assert np.allclose(pk1[m], want[1][m], rtol=ptol, atol=0)
        b._realisation = 0
        d2 = b.realise_density().double().cpu().numpy()
        assert np.array_equal(d2, dx)
        kc, pk2, err2 = b.binned_power_spectrum(nbins=nb, lognormal=True)
        assert np.allclose(pk2[m], want[1][m], rtol=ptol, atol=0)


@pytest.mark.parametrize("P,C", [(2, 4), (8, 2)])
def test_chunked_slab_transform_at_1024(P, C):
    """BASELINE config 4's size, one box in C chunks over P virtual ranks (8 points per thread, 8-column tiles: 65 tile
    columns, uneven chunks), compared with the single-GPU box through device reductions."""
    import gc
    import torch
    from fastbox_amd import CosmoBox, default_cosmo, hostgeom
    from fastbox_amd.distributed import HipSlabOps, SlabBox, run_virtual_chunked
    N, L, seed, nb = 1024, 1e3, 19, 20
    ref = CosmoBox(cosmo=default_cosmo, box_scale=L, nsamp=N, realise_now=False, precision="f32", rng="device", seed=seed)
    dx = ref.realise_density()
    want_sq, want_sum = ref.engine.sum_real(dx, squared=True), ref.engine.sum_real(dx)
    want = ref.binned_power_spectrum(delta_x=ref.lognormal(dx), nbins=nb)
    del dx
    ref.engine.close()
    gc.collect()
    boxes = [SlabBox(default_cosmo, box_scale=L, nsamp=N, precision="f32", seed=seed, rank=r, world=P,
                     ops_factory=lambda g, PP, rr: HipSlabOps(g, PP, rr, precision="f32", device=0), chunks=C)
             for r in range(P)]
    for b in boxes:
        b._pk_setup(nb, None)
    res = run_virtual_chunked(boxes, nb, True)
    got_sq = sum(float(torch.linalg.vector_norm(b.delta_x.reshape(-1), 2, dtype=torch.float64)) ** 2 for b in boxes)
    got_sum = sum(float(torch.sum(b.delta_x, dtype=torch.float64)) for b in boxes)
    assert np.isclose(got_sq, want_sq, rtol=1e-6) and abs(got_sum - want_sum) < 1e-6 * np.sqrt(want_sq * float(N) ** 3)
    h = sum(r.cpu().numpy() for r in res)
    assert hostgeom.lognormal_sums_in_range(boxes[0].ops.bin_counts(), h[0:2 * nb:2], h[1:2 * nb:2], h[2 * nb])
    mean = h[2 * nb] / float(N) ** 3
    pk, err = hostgeom.finish_bins(boxes[0].ops.bin_counts(), h[0:2 * nb:2] / mean ** 2, h[1:2 * nb:2] / mean ** 4,
                                   boxes[0].boxfactor)
    m = ~np.isnan(want[1])
    assert np.array_equal(np.isnan(pk), np.isnan(want[1])) and np.allclose(pk[m], want[1][m], rtol=1e-5, atol=0)
