#!/usr/bin/env python
"""
Headline benchmark: 512^3 box realisations per second (BASELINE.json `metric`,
configs[1]: "512^3 Gaussian box + log-normal transform + P(k) estimate on 1x MI355X").

One step = one pass of the hot path through the public API, inputs resident in HBM:

    dx = box.realise_density()                       # Philox noise, sqrt(P) colouring, c2r 3-D FFT
    pending = box.binned_power_spectrum(delta_x=box.lognormal(dx), nbins=20, wait=False)

The K pending spectra are resolved (2*20+1 doubles each) inside the timed region, after the
last step has been queued; fields never leave HBM.  Independent realisations go round-robin to
--streams boxes on their own HIP streams (default 2), so that the arithmetic-bound generator /
binning passes of one realisation run beside the memory-bound passes of another.

`python bench.py --gpus N` starts N ranks by itself (torch.distributed.run, one per GPU, RCCL) unless it
is already running under a launcher (WORLD_SIZE set).  Every rank runs independent realisations of the
same box (Monte-Carlo replicas, no data-path collective; "scaling": "weak"); `value` is the whole-job rate.

Prints ONE JSON line (see README / DESIGN.md for the fields).  Besides the contract's fields:
  roofline           dominant kernel class (strided y FFT pass), HIP events on the launch stream.  With one stream
                     the brackets sit inside the timed region (every 7th launch); with several streams kernels of
                     different boxes share the chip, so the kernel is timed in a separate single-stream pass right
                     after the timed region ("timed_in" says which)
  pipeline_roofline  whole step against 8 TB/s, on SURVEY 8(d)'s model bytes and on the bytes actually moved
  f64                the same step on a precision='f64' plan (the reference computes in complex128)
  config3            BASELINE configs[2]: gen -> v_z -> redshift space -> wedge filter -> P(k) + filtered field, independent
                     chains on --streams boxes as the headline's realisations (+ one_box, + the remap kernel's own roofline)
  roofline_gen/_bin/_z  the fused generator, binning and z passes, un-overlapped (HIP events on one stream)
  sizes              N = 1: the same step at 128^3 (BASELINE configs[0]), 256^3, 1024^3, 2048^3 on this GPU
  strong_scaling     N > 1: ONE box over all ranks (slab-decomposed FFT, RCCL all-to-all) at 1024^3 and 2048^3,
                     run as child jobs of rank 0 after the replicas leg (a failure there cannot take the line down)
  cpu_baseline       the numpy oracle (the reference's algorithm) on one host core; threaded_fft_variant: the same with scipy.fft on all cores
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nsamp", type=int, default=512)
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--nbins", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region and its roofline (no f64 / config3 / sizes / strong-scaling legs)")
    ap.add_argument("--cpu-nsamp", type=int, default=0,
                    help="grid of the CPU baseline sample; 0 = the benchmarked size itself up to 512^3 (one step, "
                         "~23 s of one host core), 256^3 scaled by voxel count above that")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "slab"],
                    help="replicas: every rank realises its own boxes (Monte-Carlo throughput, weak scaling; default). "
                         "slab: ONE box of --nsamp^3 spread over the ranks, slab-decomposed FFT with one RCCL "
                         "all-to-all per transform (strong scaling)")
    ap.add_argument("--chunks", type=int, default=None,
                    help="--mode slab: k_z chunks of ONE transform (the all-to-all of a chunk runs beside the passes of the "
                         "next); default 4 with several ranks, 1 on one rank")
    ap.add_argument("--comm", default="torch", choices=["torch", "rccl"],
                    help="--mode slab: the data path's collectives through torch.distributed (nccl backend = RCCL) or through "
                         "the library's own RCCL communicator behind the C ABI (fb_comm_create / fb_slab_exchange_begin)")
    ap.add_argument("--streams", type=int, default=2,
                    help="independent realisations are issued round-robin on this many HIP streams (boxes)")
    ap.add_argument("--sizes", default="128,256,1024,2048",
                    help="N = 1: other grid sizes of the `sizes` leg (128^3 = BASELINE configs[0], the reference's own CPU-runnable case)")
    ap.add_argument("--slab-sizes", default="1024,2048", help="N > 1: grid sizes of the `strong_scaling` leg")
    ap.add_argument("--spin-up", type=float, default=0.15,
                    help="seconds of untimed steps of the same workload run directly before the warm-up steps: the GPU "
                         "raises its clocks over ~50 ms of load (tools/step_timeline.py: the first 20 steps after an idle "
                         "gap run 10 %% below the sustained rate), and the metric is the sustained rate")
    ap.add_argument("--regions", type=int, default=5,
                    help="timed regions of --steps steps each, back to back (each bracketed by barrier + synchronize on both "
                         "sides, max over ranks); the headline is the MEDIAN region, every region's time is listed")
    ap.add_argument("--plane-batch", type=int, default=None,
                    help="x-planes per cache-resident batch of the y/z passes (default: the library's size for one box per "
                         "GPU when --streams 1, N/8 when several boxes share the GPU)")
    ap.add_argument("--plane-streams", type=int, default=None, help="1 | 2 streams for alternate plane batches")
    ap.add_argument("--no-one-box-pass", action="store_true",
                    help="skip roofline.one_box_launch (profiler runs: keeps every launch of the roofline kernel the same size)")
    ap.add_argument("--gaussian-only", action="store_true",
                    help="tuning aid: P(k) of the Gaussian field itself (no exp in the fused z pass); the line is then NOT "
                         "BASELINE's configs[1]")
    ap.add_argument("--stream-priorities", action="store_true",
                    help="tuning aid: the first box's stream gets the device's highest priority, the others the lowest")
    ap.add_argument("--kernel-event-stride", type=int, default=7,
                    help="--streams 1: bracket every K-th launch of the roofline kernel with a HIP event pair inside "
                         "the timed region (an event pair costs ~3 us of stream time)")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="--streams 1: bracket every kernel with HIP events (per-kernel breakdown; costs ~4 %% of the rate)")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_LAUNCH_ENV = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
               "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID",
               "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE",
               "TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCHELASTIC_ERROR_FILE", "OMP_NUM_THREADS")


def spawn_ranks(nranks, extra_args, timeout=None):
    """Run this script as `nranks` fresh processes (torch.distributed.run on 127.0.0.1, one rank per GPU) and return
    (return code, stdout).  Children are new processes started with subprocess: nothing that has touched the GPU is
    re-executed.  The launcher's own environment variables are removed so that a nested job forms its own group."""
    env = {k: v for k, v in os.environ.items() if k not in _LAUNCH_ENV}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(extra_args)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, start_new_session=True, text=True)
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, 15)
            time.sleep(5)
            os.killpg(proc.pid, 9)
        except OSError:
            pass
        out, _ = proc.communicate()
        return -9, out
    return proc.returncode, out


def _last_json(text):
    for ln in reversed((text or "").strip().splitlines()):
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def cpu_baseline(nsamp_bench, nsamp_cpu, nbins):
    """The reference algorithm (oracle restatement: numpy pocketfft, 1 thread) on the host,
    on a bounded sample, scaled to the metric's unit by voxel count."""
    import numpy as np
    from oracle import box_oracle as bo
    from oracle import standin
    geo = bo.box_geometry(1e3, nsamp_cpu)
    cosmo = standin.cosmology()
    rng = np.random.RandomState(1)
    t0 = time.time()
    re, im = bo.draw_noise(nsamp_cpu, rng)
    dx, dk = bo.realise_density(geo, standin.pk_fn(cosmo, 1.0), re, im)
    ln = bo.lognormal(dx)
    bo.binned_power_spectrum(geo, np.fft.fftn(ln), nbins=nbins)
    dt = time.time() - t0
    scale = (nsamp_bench / float(nsamp_cpu)) ** 3
    how = "the workload's own size, not scaled" if nsamp_cpu == nsamp_bench else \
        "scaled by voxel count x%.0f to %d^3" % (scale, nsamp_bench)
    out = {"value": 1.0 / (dt * scale), "unit": "boxes/s", "cores": 1, "kind": "port",
           "sample": "one %d^3 realise_density + lognormal + binned_power_spectrum with the numpy oracle "
                     "(%.1f s, 1 thread), %s" % (nsamp_cpu, dt, how)}
    try:        # SURVEY 8d's best-effort CPU line: the same step with its four 3-D FFTs on every core this job may use
        cores = max(1, min(len(os.sched_getaffinity(0)), 16))
        rng = np.random.RandomState(1)
        with bo.threaded_fft(cores) as tf:
            t0 = time.time()
            re, im = bo.draw_noise(nsamp_cpu, rng)
            dx, dk = bo.realise_density(geo, standin.pk_fn(cosmo, 1.0), re, im)
            ln = bo.lognormal(dx)
            bo.binned_power_spectrum(geo, tf.fftn(ln), nbins=nbins)
            dtt = time.time() - t0
        out["threaded_fft_variant"] = {
            "value": 1.0 / (dtt * scale), "unit": "boxes/s", "cores": cores,
            "sample": "the same step with scipy.fft on %d threads (%.1f s): the reference's random draws and its binning "
                      "loop stay on one core, as in fastbox/box.py" % (cores, dtt)}
    except Exception as e:
        out["threaded_fft_variant"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def _git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except Exception:
        return None


def from_profiles(N, precision):
    """What the committed rocprofv3 summaries (profiles/, collected off-line by tools/collect_profiles.sh over this
    command) say about the roofline kernel: PMC traffic per launch (FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md) and the --kernel-trace --stats average duration.  Labelled with the file and the commit it
    was collected at: these are NOT measured by this run."""
    out = {"file": None, "head": None, "traffic_bytes_per_launch": None, "rocprof_avg_launch_us": None}
    try:
        idx = json.load(open(os.path.join(ROOT, "profiles", "current.json")))
        out["head"] = idx.get("head")
        out["file"] = [idx.get("pmc"), idx.get("kernel_stats")]
        tname = "float" if precision == "f32" else "double"
        pmc = json.load(open(os.path.join(ROOT, "profiles", idx["pmc"])))
        key = [k for k in pmc if "k_fft_strided<%s, %d, 0" % (tname, N) in k]
        if key:
            c = pmc[key[0]]
            out["traffic_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024.0
        import csv
        with open(os.path.join(ROOT, "profiles", idx["kernel_stats"])) as fh:
            for row in csv.DictReader(fh):
                if ("k_fft_strided<%s, %d, 0" % (tname, N)) in row["Name"]:
                    out["rocprof_avg_launch_us"] = float(row["AverageNs"]) * 1e-3
    except Exception:
        pass
    return out


def _collective_info(torch, dist, local_rank):
    """What the process group of this job really was (every rank calls it): backend, the world size the group reports and,
    all-gathered, each rank's device -- so that a scaling record shows N ranks on N different GPUs behind RCCL."""
    me = {"rank": dist.get_rank(), "host": socket.gethostname(), "device": local_rank, "name": None, "pci_bus_id": None, "uuid": None}
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        me["name"] = pr.name
        dom, bus, dev = getattr(pr, "pci_domain_id", None), getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None)
        if bus is not None:
            me["pci_bus_id"] = "%04x:%02x:%02x" % (dom or 0, bus, dev or 0)
        me["uuid"] = str(getattr(pr, "uuid", None))
    except Exception as e:
        me["error"] = "%s: %s" % (type(e).__name__, e)
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, me)
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "devices": gathered,
            "distinct_devices": len({(g["host"], g["pci_bus_id"] or g["device"]) for g in gathered})}


def _prio(args, i):
    """--stream-priorities: box 0 on the highest-priority stream, the others on the lowest (tuning aid)."""
    if not getattr(args, "stream_priorities", False):
        return None
    return -1 if i == 0 else 1


def _make_boxes(args, N, precision, n, rank, local_rank):
    from fastbox_amd import CosmoBox, default_cosmo
    from fastbox_amd.device import new_stream
    boxes = [CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, redshift=0., realise_now=False,
                      precision=precision, rng="device", seed=1000 * (rank + 1) + i, device=local_rank,
                      stream=(new_stream(local_rank, _prio(args, i)) if n > 1 else None)) for i in range(max(1, n))]
    # several boxes share the GPU's 256 MiB Infinity Cache: each keeps a smaller plane batch resident
    # (512^3 fp32, two boxes: 64 planes = 2 x 70 MB of half spectrum + real planes; measured in profiles/r02_*)
    sz = 4 if precision == "f32" else 8
    plane_bytes = (N + 1) * ((N // 2 + 16) // 16 * 16) * 2 * sz + N * N * sz       # half-spectrum plane + real plane
    # (2048^3, single precision: the y passes run in 128-byte rows, 64 tiles per plane, and want >= 16 planes per launch:
    # 12.1 boxes/s against 11.8 with 4, profiles/r04_pass_bench_2048.txt)
    share = max(16 if (N >= 2048 and precision == "f32") else 4, int(280e6 / max(1, n) / plane_bytes))
    pb = args.plane_batch if args.plane_batch is not None else (-1 if n == 1 or share >= N else share)
    ps = args.plane_streams if args.plane_streams is not None else (0 if n == 1 else 1)
    for b in boxes:
        b.engine.set_plane_batching(pb, ps)
        b._bench_plane_batching = (pb, ps)
    return boxes


def _step_fn(boxes, nbins, lognormal=True, keep_field=True):
    counter = [0]

    def step():
        box = boxes[counter[0] % len(boxes)]
        counter[0] += 1
        dx = box.realise_density()
        return box.binned_power_spectrum(delta_x=box.lognormal(dx) if lognormal else dx, nbins=nbins, wait=False,
                                         keep_field=keep_field)
    return step


def _check_finite(triples, what):
    """Every (kc, pk, stddev) of a leg: finite on the bins that hold modes, the same empty-bin mask throughout.
    A leg that timed garbage raises instead of reporting a rate; returns the number of spectra checked."""
    import numpy as np
    mask = None
    for kc, pk, err in triples:
        empty = np.isnan(pk)
        if mask is None:
            mask = empty
        if not np.array_equal(empty, mask) or not np.all(np.isfinite(pk[~empty])) or not np.all(np.isfinite(err[~empty])) \
                or not np.all(np.isfinite(kc)) or empty.all():
            raise FloatingPointError("%s: non-finite power spectrum %r" % (what, pk))
    return len(triples)


def _warm(step, n):
    """n untimed steps, queued back to back like the timed ones (the GPU stays busy up to the fence)."""
    for p in [step() for _ in range(n)]:
        p.result()


def _timed_steps(step, steps, warmup, sync, spin_s=0.1, what="leg"):
    # the legs start from an idle GPU (plans were just created): untimed steps until the clocks are up
    # (profiles/r02_step_timeline.txt: ~40-60 steps of the 512^3 workload after an idle gap)
    t_spin = time.perf_counter()
    _warm(step, warmup)
    while time.perf_counter() - t_spin < spin_s:
        _warm(step, max(2, warmup))
    sync()
    t0 = time.perf_counter()
    pend = [step() for _ in range(steps)]
    out = [p.result() for p in pend]
    sync()
    dt = time.perf_counter() - t0
    _check_finite(out, what)
    return dt


def quick_rate(args, N, precision, steps, warmup, rank, local_rank, torch, keep_field=True):
    """boxes/s of the benchmarked step at another size / precision (short run, same streams setting)."""
    boxes = _make_boxes(args, N, precision, args.streams, rank, local_rank)
    try:
        dt = _timed_steps(_step_fn(boxes, args.nbins, keep_field=keep_field), steps, max(warmup, len(boxes)),
                          torch.cuda.synchronize, what="%d^3 %s" % (N, precision))
        repeats = sum(getattr(b, "lognormal_repeats", 0) for b in boxes)
    finally:
        for b in boxes:
            b.engine.close()
    s = 4 if precision == "f32" else 8
    sweep = float(N) ** 3 * 2 * s
    rate = steps / dt
    return {"nsamp": N, "dtype": precision, "value": rate, "unit": "boxes/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "pipeline_frac_model_bytes": 5.0 * sweep * rate / 1e9 / HBM_PEAK_GBS,
            "finite": True, "lognormal_repeats": repeats}       # (a non-finite spectrum raises: _check_finite)


def config3_leg(args, N, rank, local_rank, torch, chains=30):
    """BASELINE configs[2] on one GPU: gen -> v_z -> redshift-space remap -> k_perp/k_par wedge filter -> P(k) of the
    filtered field + the filtered field itself, resident in HBM (13.5 sweeps).  Independent chains
    round-robin on --streams boxes, each on its own HIP stream, as the headline's realisations; the one-box rate and
    the remap kernel's own roofline (un-overlapped, every launch bracketed) are measured on the first box alone."""
    from fastbox_amd import Wedge
    from fastbox_amd.box import RedshiftSpaceField
    boxes = _make_boxes(args, N, "f32", max(1, args.streams), rank, local_rank)
    wedge = Wedge(slope=0.3)
    fused = [False]

    def chain_on(box):
        dx = box.realise_density()
        vz = box.to_real(box.realise_velocity()[2])
        ds = box.redshift_space_density(delta_x=dx, velocity_z=vz, sigma_nl=0.0)
        fused[0] = isinstance(ds, RedshiftSpaceField)
        filt = box.apply_transfer_fn(box.to_k(ds), wedge)
        pk = box.binned_power_spectrum(delta_x=filt.real, nbins=20, wait=False)
        filt.ptr                                                # deliver the filtered field as well
        return pk

    def timed(use, n):
        count = [0]

        def chain():
            count[0] += 1
            return chain_on(use[count[0] % len(use)])
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.1:               # from an idle GPU: untimed chains until the clocks are up
            for p in [chain() for _ in range(3 * len(use))]:
                p.result()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = [chain() for _ in range(n)]
        out = [p.result() for p in pend]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        _check_finite(out, "config3")
        return dt
    try:
        # (the two configurations alternately, twice: the first timed region of a leg still sees the clocks settle)
        one = boxes[:1]
        shared = getattr(boxes[0], "_bench_plane_batching", (-1, 0))
        dts, dts_one = [], []
        for _ in range(2):
            boxes[0].engine.set_plane_batching(*shared)
            dts.append(timed(boxes, chains))
            if len(boxes) > 1:
                boxes[0].engine.set_plane_batching(-1, 0)       # alone on the GPU: the library's own batching
                dts_one.append(timed(one, chains))
        dt = min(dts)
        dt_one = min(dts_one) if dts_one else dt
        # the remap kernel alone, un-overlapped: HIP events around every launch of its class
        eng = boxes[0].engine
        eng.profile_start(["rsd"], stride=1)
        pend = [chain_on(boxes[0]) for _ in range(chains)]
        for p in pend:
            p.result()
        prof = eng.profile_stop()
    finally:
        for b in boxes:
            b.engine.close()
    sweep = float(N) ** 3 * 8
    rsd_ms, rsd_n = prof["rsd"]
    # separate kernel: R delta 1/2 + R v_z 1/2 + W 1/2 sweeps.  Fused z pass (k_rsd_turn: both inverse z transforms, the
    # remap, the forward z transform): R 2 x 1/2 (the two work spectra) + W 1/2 (delta_x) + W 1/2 (forward z spectrum)
    rsd_bytes = (2.0 if fused[0] else 1.5) * sweep
    per_chain_ms = rsd_ms / chains
    rsd_gbs = rsd_bytes / (per_chain_ms * 1e-3) / 1e9 if rsd_ms > 0 else None
    return {"workload": "%d^3: realise_density -> realise_velocity[2] -> redshift_space_density -> apply_transfer_fn(Wedge "
                        "slope 0.3) -> binned_power_spectrum + filtered field; value = best of {1, %d} boxes per GPU (both listed)"
                        % (N, len(boxes)),
            # the better of the two ways to run independent chains on one GPU is the leg's value; both are listed
            "ms_per_chain": 1e3 * min(dt, dt_one), "value": 1.0 / min(dt, dt_one), "unit": "chains/s", "chains": chains, "dtype": "f32",
            "streams_per_gpu": len(boxes) if dt <= dt_one else 1,
            "boxes_%d" % len(boxes): {"ms_per_chain": 1e3 * dt, "value": 1.0 / dt, "unit": "chains/s"},
            "one_box": {"ms_per_chain": 1e3 * dt_one, "value": 1.0 / dt_one, "unit": "chains/s"},
            "model_sweeps": 13.5, "pipeline_frac_model_bytes": 13.5 * sweep / min(dt, dt_one) / 1e9 / HBM_PEAK_GBS, "finite": True,
            "rsd_roofline": {"kernel": "k_rsd_turn (c2r of delta and v_z + line-of-sight remap + r2c, one kernel)" if fused[0]
                                       else "k_rsd_cells",
                             "bound": "hbm", "algorithmic_bytes": rsd_bytes, "ms_per_chain": per_chain_ms,
                             "launches_per_chain": rsd_n / float(chains), "achieved": rsd_gbs, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": rsd_gbs / HBM_PEAK_GBS if rsd_gbs else None,
                             "note": "the kernel is bound by LDS and issue slots, not by memory (DESIGN.md 5, round 3): its "
                                     "2 sweeps replace the 4.5 of the three z passes and the remap kernel it stands for"
                                     if fused[0] else None}}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand as `python bench.py --gpus N`: become the launcher.  Nothing has touched the GPU yet.
        rc, out = spawn_ranks(args.gpus, sys.argv[1:])
        sys.stdout.write(out or "")
        sys.stdout.flush()
        sys.exit(rc if rc >= 0 else 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch                                  # first: pins one HIP runtime for the process
    import torch.distributed as dist
    import numpy as np
    # rehearsal hooks (not used by the driver): FASTBOX_BENCH_BACKEND=gloo runs the multi-process control
    # flow without RCCL, FASTBOX_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 of a single-GPU box
    backend = os.environ.get("FASTBOX_BENCH_BACKEND", "nccl")
    if os.environ.get("FASTBOX_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    args._reduce_device = "cuda" if backend == "nccl" else "cpu"

    N = args.nsamp
    if args.mode == "slab":
        return slab_main(args, rank, world, local_rank, torch, dist, np)
    nstreams = max(1, args.streams)
    boxes = _make_boxes(args, N, args.precision, nstreams, rank, local_rank)
    step = _step_fn(boxes, args.nbins, not args.gaussian_only)
    eng = boxes[0].engine

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The auxiliary legs run FIRST (rank 0): they are measurements of their own, and they leave the GPU at its
    # sustained clocks, so that the timed region below is the steady state SURVEY 8(d) defines the metric on
    # (a cold start reads ~4 % low over a 20-step region: tools/step_timeline.py, profiles/r02_step_timeline.txt).
    extras = {}
    if rank == 0 and not args.no_extras:
        def guarded(name, fn):
            try:
                extras[name] = fn()
            except Exception as e:                     # an extra leg must never take the headline down
                extras[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        other = "f64" if args.precision == "f32" else "f32"
        guarded("config3", lambda: config3_leg(args, N, rank, local_rank, torch))

        def spectra_only():
            r = quick_rate(args, N, args.precision, 100, 10, rank, local_rank, torch, keep_field=False)
            r["note"] = "NOT the headline: the same step with binned_power_spectrum(..., keep_field=False) -- delta_x is not " \
                        "written by the fused z pass (4.0 instead of 4.5 sweeps moved; a Monte-Carlo covariance run, " \
                        "fastbox_amd/montecarlo.py, needs the spectra only and can redraw any realisation by index)"
            return r
        guarded("spectra_only_variant", spectra_only)
        guarded(other, lambda: quick_rate(args, N, other, 80, 20, rank, local_rank, torch))
        if world == 1:
            def sizes():
                out = {}
                for n2 in [int(x) for x in args.sizes.split(",") if x]:
                    if n2 != N:
                        st = max(8, min(100, int(100 * (256. / n2) ** 3)))     # >= 4 steps per box of a two-box run
                        out[str(n2)] = quick_rate(args, n2, args.precision, st, 4, rank, local_rank, torch)
                return out
            guarded("sizes", sizes)
    if world > 1:
        dist.barrier()                    # the other ranks wait HERE for rank 0's legs, so that every rank spins its GPU up
                                          # directly before the timed region (an idle GPU lowers its clocks again)
    in_region = nstreams == 1             # event brackets inside the timed region only when kernels run alone
    ev_stride = 1 if args.all_kernel_events else max(1, args.kernel_event_stride)
    prof_steps = args.steps
    if not in_region:
        # kernels of different boxes overlap in the timed region, so the roofline kernel is timed on its own here, on
        # ONE stream, directly before the warm-up steps (this pass also leaves the GPU at its sustained clocks)
        prof_steps = 40
        one = _step_fn(boxes[:1], args.nbins, not args.gaussian_only)
        one().result()
        torch.cuda.synchronize()
        eng.profile_start(["fft_strided"], stride=1)
        for p in [one() for _ in range(prof_steps)]:
            p.result()
        prof = eng.profile_stop()
        plain_launches = prof["fft_strided"][1]
        # the other kernels of the step in the same un-overlapped way (one stream, every launch bracketed): the fused
        # generator pass, the fused binning pass and the fused z pass -- the two that sit farthest from the roofline
        # belong in this line, not only in profiles/
        eng.profile_start(["fft_gen", "fft_bin", "fft_contig"], stride=1)
        for p in [one() for _ in range(prof_steps)]:
            p.result()
        prof_other = eng.profile_stop()
        # the same kernel in the launches one box per GPU would use (plane batches sized to the whole Infinity Cache:
        # a launch carries a fixed fill and drain, so the larger launch reads closer to the kernel's own rate)
        big = None
        try:
            if args.no_one_box_pass:
                raise StopIteration
            eng.set_plane_batching(-1, 0)
            one().result()
            torch.cuda.synchronize()
            eng.profile_start(["fft_strided"], stride=1)
            for p in [one() for _ in range(20)]:
                p.result()
            pb_ = eng.profile_stop()["fft_strided"]
            big = (pb_[0], pb_[1], 20)
        except StopIteration:
            pass
        finally:
            eng.set_plane_batching(*boxes[0]._bench_plane_batching)
            one().result()
    import gc
    gc.collect()                          # BEFORE the spin-up: an idle gap directly before the timed region would let the
    gc.disable()                          # clocks drop again; a collector pause inside a 20 ms region would be 5 % of it
    t_spin = time.perf_counter()
    spin_steps = 0
    while time.perf_counter() - t_spin < args.spin_up:
        _warm(step, 10)
        spin_steps += 10
    _warm(step, max(args.warmup, len(boxes)))
    fence()
    if in_region:
        eng.profile_start(None if args.all_kernel_events else ["fft_strided"], stride=ev_stride)
    # the metric is the sustained rate -- the median of `regions` back-to-back regions of exactly K steps, each
    # fenced (barrier + synchronize) on both sides; `steps` / `ms_per_step` describe ONE region (the median one)
    nreg = max(1, args.regions)
    dts, spectra = [], []
    for _ in range(nreg):
        t0 = time.perf_counter()
        pending = [step() for _ in range(args.steps)]
        spectra += [pnd.result() for pnd in pending]
        if in_region and len(dts) == nreg - 1:
            prof = eng.profile_stop()                    # synchronises the launch stream
            plain_launches = eng.profile_seen() if ev_stride > 1 else prof["fft_strided"][1]
        fence()
        dts.append(time.perf_counter() - t0)
    if in_region:
        prof_steps = args.steps * nreg
    gc.enable()
    _check_finite(spectra, "timed region")       # every rank: a rank that timed garbage stops the job
    ln_repeats = sum(getattr(b, "lognormal_repeats", 0) for b in boxes)
    collective = None
    if world > 1:
        t = torch.tensor(dts, dtype=torch.float64, device=args._reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)         # per region: the slowest rank's time
        dts = [float(x) for x in t.tolist()]
        collective = _collective_info(torch, dist, local_rank)
        dist.barrier()
        dist.destroy_process_group()
    dt = sorted(dts)[len(dts) // 2]
    if rank != 0:
        return

    # ---------------------------------------------------------------- rank 0: the line
    s = 4 if args.precision == "f32" else 8
    timed_in = "timed region (every %d%s launch bracketed)" % (ev_stride, "th" if ev_stride > 1 else "st")
    if not in_region:
        timed_in = "separate pass of %d steps on ONE stream directly before the warm-up steps (in the timed region itself " \
                   "kernels of %d boxes share the chip); every launch bracketed" % (prof_steps, nstreams)
    ms, launches = prof["fft_strided"]
    # a step holds two strided y passes (inverse and forward), each launched once per x-plane batch (fb_fft_launch.inc
    # yz_passes): one launch reads and writes its share of N*N*(N/2) complex values (packed work spectrum)
    ncols = N // 2
    alg_bytes = 2 * (2.0 * N * N * ncols * 2 * s) * prof_steps / max(plain_launches, 1)
    achieved = alg_bytes / (ms / max(launches, 1) * 1e-3) / 1e9 if ms > 0 else None
    fp = from_profiles(N, args.precision)
    line = {
        "metric": "%d^3 box realisations/sec (gen + log-normal + P(k))" % N if not args.gaussian_only
                  else "%d^3 box realisations/sec (gen + P(k), NO log-normal: tuning run)" % N,
        "value": world * args.steps / dt, "unit": "boxes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "finite": True, "lognormal_repeats": ln_repeats,
        "regions": {"count": len(dts), "ms_per_step": [round(1e3 * d / args.steps, 5) for d in dts],
                    "boxes_per_s": [round(world * args.steps / d, 2) for d in dts],
                    "spread": (max(dts) - min(dts)) / dt,
                    "what": "back-to-back timed regions of `steps` steps, each fenced on both sides (max over ranks per "
                            "region); value / ms_per_step are the MEDIAN region's"},
        "collective": collective,
        "config": {"workload": "%d^3 Gaussian box (device Philox4x32-10 noise, stand-in EH P(k), L=1000 Mpc) + "
                               "log-normal transform + binned P(k), nbins=%d" % (N, args.nbins),
                   "nsamp": N, "parallelism": "replicas x%d" % world, "streams_per_gpu": len(boxes),
                   "warmup_steps_run": max(args.warmup, len(boxes)), "spin_up_s": args.spin_up,
                   "spin_up_steps": spin_steps},
        "roofline": {"bound": "hbm", "kernel": "k_fft_strided (y FFT pass over a plane batch of the half spectrum)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS if achieved else None,
                     "traffic": fp["traffic_bytes_per_launch"], "algorithmic_bytes": alg_bytes,
                     "avg_launch_us": 1e3 * ms / max(launches, 1), "launches": plain_launches,
                     "launches_timed": launches, "timed_in": timed_in,
                     "note": "plane batches are sized to the 256 MiB Infinity Cache, so most of these bytes are served "
                             "on-die: this is L2 <-> fabric bandwidth; the whole-step figure (pipeline_roofline) is the HBM statement"},
        "from_profiles": fp,
    }
    if not in_region:
        half = float(N) * N * (N // 2) * 2 * s                      # one sweep of the packed work spectrum = of the real field
        for key, cls, nbytes, what in (
                ("roofline_gen", "fft_gen", half, "k_fft_strided<GEN>: Philox noise + sqrt(P) colouring + first inverse pass (x); writes the work spectrum"),
                ("roofline_bin", "fft_bin", half, "k_fft_strided<BIN>: last forward pass (x) + |delta_k|^2 shell binning; reads the work spectrum"),
                ("roofline_z", "fft_contig", 3 * half, "k_fft_contig<C2R2C>: inverse z pass, delta_x stored, exp(), forward z pass (per plane batch: "
                                                       "reads 1/2 sweep, writes 2 x 1/2)")):
            ms_c, n_c = prof_other.get(cls, (0.0, 0))
            if n_c and ms_c > 0:
                per_step = n_c / float(prof_steps)                  # launches per step (the z pass: one per plane batch)
                us = 1e3 * ms_c / n_c
                ach = nbytes / per_step / (us * 1e-6) / 1e9
                line[key] = {"bound": "hbm", "kernel": what, "algorithmic_bytes": nbytes / per_step, "avg_launch_us": us,
                             "launches_timed": n_c, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach / HBM_PEAK_GBS, "timed_in": "separate pass of %d steps on ONE stream before the "
                             "warm-up steps, every launch bracketed with HIP events" % prof_steps}
    if not in_region and big and big[1]:
        b_alg = 2 * (2.0 * N * N * ncols * 2 * s) * big[2] / big[1]
        b_ach = b_alg / (big[0] / big[1] * 1e-3) / 1e9
        line["roofline"]["one_box_launch"] = {
            "what": "the same kernel, same un-overlapped method, in the larger launches one box per GPU uses (plane "
                    "batches sized to the whole Infinity Cache)",
            "algorithmic_bytes": b_alg, "avg_launch_us": 1e3 * big[0] / big[1], "launches_timed": big[1],
            "achieved": b_ach, "frac": b_ach / HBM_PEAK_GBS}
    if in_region:
        total_ms = sum(v[0] for v in prof.values()) * (plain_launches / max(launches, 1) if ev_stride > 1 else 1.)
        line["kernel_ms_per_step"] = {k: round(v[0] / prof_steps * (plain_launches / max(v[1], 1) if ev_stride > 1 else 1.), 4)
                                      for k, v in prof.items() if v[1]}
        line["kernel_ms_total_per_step"] = round(total_ms / prof_steps, 4)
    # whole step against the HBM roofline: SURVEY 8(d)'s byte model for this workload is 5.0 sweeps of N^3 complex
    # values; this implementation moves 4.5 (the z passes of realisation and estimate are one), of N*N*(N/2) columns
    sweep = float(N) ** 3 * 2 * s
    boxes_per_s = line["value"] / world            # per GPU
    moved = 4.5 * sweep
    line["pipeline_roofline"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "model_sweeps": 5.0, "model_bytes_per_box": 5.0 * sweep,
                                 "achieved": 5.0 * sweep * boxes_per_s / 1e9,
                                 "frac": 5.0 * sweep * boxes_per_s / 1e9 / HBM_PEAK_GBS,
                                 "moved_sweeps": 4.5, "moved_bytes_per_box": moved,
                                 "achieved_moved": moved * boxes_per_s / 1e9,
                                 "frac_moved": moved * boxes_per_s / 1e9 / HBM_PEAK_GBS}
    for b in boxes:
        b.engine.close()
    del boxes
    line.update(extras)
    if not args.no_extras and world > 1:
        # one box over all ranks, as child jobs (the other ranks of this job have left their GPUs by now)
        out = {}
        torch.cuda.empty_cache()
        failed = False
        for n2 in [int(x) for x in args.slab_sizes.split(",") if x]:
            st = 10 if n2 <= 1024 else 5
            if failed:                                  # do not spend another time-out on a transport that does not work
                out[str(n2)] = {"error": "skipped: the previous slab job failed"}
                continue
            try:
                rc, txt = spawn_ranks(world, ["--mode", "slab", "--nsamp", str(n2), "--steps", str(st), "--warmup", "2",
                                              "--gpus", str(world), "--precision", args.precision, "--chunks", "4"], timeout=240)
                j = _last_json(txt)
                out[str(n2)] = j if (rc == 0 and j) else {"error": "slab job rc=%s" % rc}
                failed = not (rc == 0 and j)
                if not failed and n2 == min(int(x) for x in args.slab_sizes.split(",") if x) \
                        and os.environ.get("FASTBOX_BENCH_BACKEND", "nccl") == "nccl":       # (not in the one-device rehearsal)
                    # the same job with the collectives behind the C ABI (the library's own RCCL communicator), smallest size
                    rc, txt = spawn_ranks(world, ["--mode", "slab", "--nsamp", str(n2), "--steps", str(st), "--warmup", "2",
                                                  "--gpus", str(world), "--precision", args.precision, "--chunks", "4",
                                                  "--comm", "rccl"], timeout=240)
                    j = _last_json(txt)
                    out["%d_comm_rccl_abi" % n2] = j if (rc == 0 and j) else {"error": "slab job (--comm rccl) rc=%s" % rc}
            except Exception as e:
                out[str(n2)] = {"error": "%s: %s" % (type(e).__name__, e)}
                failed = True
        line["strong_scaling"] = out
    if not args.no_cpu_baseline and world == 1:       # rank 0 at N = 1 only
        line["cpu_baseline"] = cpu_baseline(N, args.cpu_nsamp or (N if N <= 512 else 256), args.nbins)
    else:
        line["cpu_baseline"] = None
    line["head"] = _git_head()
    print(json.dumps(line))


def slab_main(args, rank, world, local_rank, torch, dist, np):
    """One box over all ranks: SlabBox (x-slabs <-> k_y-slabs, one all-to-all per transform)."""
    from fastbox_amd import default_cosmo
    from fastbox_amd.distributed import SlabBox
    N = args.nsamp
    torch.cuda.set_device(local_rank)
    chunks = args.chunks if args.chunks is not None else (4 if world > 1 else 1)
    comm = "rccl" if args.comm == "rccl" else None
    box = SlabBox(default_cosmo, box_scale=1e3, nsamp=N, precision=args.precision, seed=1000, rank=rank, world=world,
                  device=local_rank, chunks=chunks, comm=comm)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(nsteps, wait):
        """wait=False: Monte-Carlo steps pipelined (exchanges of one realisation behind the passes of its neighbours);
        wait=True: one realisation at a time, every exchange exposed."""
        fence()
        t0 = time.perf_counter()
        if wait:
            out = [box.realise_and_power(nbins=args.nbins, lognormal=True) for _ in range(nsteps)]
        else:
            tickets = [box.realise_and_power(nbins=args.nbins, lognormal=True, wait=False) for _ in range(nsteps)]
            out = [tk.result() for tk in tickets]
        fence()
        d = time.perf_counter() - t0
        _check_finite(out, "slab %d^3" % N)
        if world > 1:
            t = torch.tensor([d], dtype=torch.float64, device=args._reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d

    collective = _collective_info(torch, dist, local_rank) if world > 1 else None
    if collective is not None:
        collective["data_path"] = "torch.distributed all_to_all_single / all_reduce" if comm is None else \
            "libfastbox_hip.so: fb_slab_exchange_begin / _wait (ncclSend + ncclRecv group on the library's stream), fb_allreduce_f64"
        if comm is not None:
            collective["library_comm"] = box._comm.info()
    for _ in range(args.warmup):
        box.realise_and_power(nbins=args.nbins, lognormal=True)
    timed(3, False)
    dt = timed(args.steps, False)
    # one realisation at a time (BASELINE config 4: ONE box): with chunks > 1 each all-to-all runs beside the passes of
    # the neighbouring chunks of the same transform; on one rank the figure is the cost of the chunked form itself
    dt_sync = timed(max(2, args.steps // 2), True) / max(2, args.steps // 2) if (world > 1 or box.chunks > 1) else None
    # the same one-at-a-time loop with the transform in ONE piece (every all-to-all fully exposed): what the chunks buy
    dt_sync1 = None
    if world > 1 and box.chunks > 1:
        if comm is not None:
            os.environ["FASTBOX_RDV_OFFSET"] = "23"          # a second communicator: its id travels on another port
        box1 = SlabBox(default_cosmo, box_scale=1e3, nsamp=N, precision=args.precision, seed=1000, rank=rank, world=world,
                       device=local_rank, chunks=1, comm=comm)
        keep, box = box, box1
        try:
            box.realise_and_power(nbins=args.nbins, lognormal=True)
            dt_sync1 = timed(max(2, args.steps // 2), True) / max(2, args.steps // 2)
        finally:
            box = keep
    if rank == 0:
        s = 4 if args.precision == "f32" else 8
        sweep = float(N) ** 3 * 2 * s
        a2a = (N // world) ** 2 * ((N // 2 + 16) // 16 * 16) * 2 * s
        timing = None if dt_sync is None else {
            "ms_per_step_one_realisation_at_a_time": 1e3 * dt_sync, "chunks_per_transform": box.chunks,
            "ms_per_step_pipelined": 1e3 * dt / args.steps,
            "ms_per_step_one_realisation_at_a_time_unchunked": None if dt_sync1 is None else 1e3 * dt_sync1,
            "exposed_exchange_ms_per_step": None if dt_sync is None else 1e3 * (dt_sync - dt / args.steps),
            "all_to_alls_per_step": 2, "bytes_sent_per_rank_per_all_to_all": a2a * (world - 1),
            "note": "pipelined = up to three realisations in flight, both all-to-alls asynchronous on the RCCL stream "
                    "behind other realisations' passes (several ranks; on one rank it is the same loop as the first "
                    "figure); one realisation at a time = every transform in `chunks_per_transform` k_z chunks, the "
                    "all-to-all of a chunk beside the passes of the next"}
        print(json.dumps({
            "metric": "%d^3 box realisations/sec (gen + log-normal + P(k)), one box over all GPUs" % N,
            "value": args.steps / dt, "unit": "boxes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic", "finite": True, "lognormal_repeats": box.ln_repeats,
            "config": {"workload": "%d^3 Gaussian box + log-normal + binned P(k), slab-decomposed FFT, one all-to-all "
                                   "per transform" % N, "nsamp": N,
                       "parallelism": "slab x%d" % world,
                       "all_to_all_bytes_per_rank_pair": (N // world) ** 2 * ((N // 2 + 16) // 16 * 16) * 2 * s},
            "pipeline_frac_model_bytes": 5.0 * sweep * (args.steps / dt) / 1e9 / (HBM_PEAK_GBS * world),
            "exchange": timing, "collective": collective, "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
