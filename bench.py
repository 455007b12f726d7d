#!/usr/bin/env python
"""
Headline benchmark: 512^3 box realisations per second (BASELINE.json `metric`,
configs[1]: "512^3 Gaussian box + log-normal transform + P(k) estimate on 1x MI355X").

One step = one pass of the hot path through the public API, inputs resident in HBM:

    dx = box.realise_density()                       # Threefry noise, sqrt(P) colouring, c2r 3-D FFT
    pending = box.binned_power_spectrum(delta_x=box.lognormal(dx), nbins=20, wait=False)

The K pending spectra are resolved (2*20+1 doubles each) inside the timed region, after the
last step has been queued; fields never leave HBM.  With --gpus N every rank runs
independent realisations of the same box (Monte-Carlo replicas, no data-path collective;
"scaling": "weak"); `value` is the whole-job rate.

Prints ONE JSON line (see README / DESIGN.md for the fields).  `roofline` is for the
dominant kernel class (the strided x/y FFT passes), timed live with HIP events on the
launch stream inside the timed region -- every 7th of its launches is bracketed (`launches_timed`
of `launches`; --kernel-event-stride), since an event pair per launch costs 4 % of the rate
without changing the average; `cpu_baseline` is the numpy oracle (the reference's algorithm)
on one core.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nsamp", type=int, default=512)
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--nbins", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-nsamp", type=int, default=0,
                    help="grid of the CPU baseline sample; 0 = the benchmarked size itself up to 512^3 (one step, "
                         "~23 s of one host core), 256^3 scaled by voxel count above that")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "slab"],
                    help="replicas: every rank realises its own boxes (Monte-Carlo throughput, weak scaling; default). "
                         "slab: ONE box of --nsamp^3 spread over the ranks, slab-decomposed FFT with one RCCL "
                         "all-to-all per transform (strong scaling)")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent realisations are issued round-robin on this many HIP streams (boxes), so that "
                         "the compute-bound passes of one overlap the HBM-bound passes of the next (+8 %% at 2; "
                         "default 1 so that per-kernel durations, and the roofline derived from them, are those of "
                         "kernels running alone)")
    ap.add_argument("--kernel-event-stride", type=int, default=7,
                    help="bracket every K-th launch of the roofline kernel with a HIP event pair, inside the timed region "
                         "(K coprime to the 8 launches per step, so every position of the step is sampled equally); an "
                         "event pair costs ~3 us of stream time, so K = 1 (every launch) lowers the measured rate by 4 %% "
                         "while the average launch duration comes out the same (51.1 vs 51.4 us)")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="bracket every kernel with HIP events (per-kernel breakdown; costs ~4 %% of the rate); "
                         "by default only the dominant kernel class is bracketed")
    return ap.parse_args()


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
    return {"value": 1.0 / (dt * scale), "unit": "boxes/s", "cores": 1, "kind": "port",
            "sample": "one %d^3 realise_density + lognormal + binned_power_spectrum with the numpy oracle "
                      "(%.1f s, 1 thread), %s" % (nsamp_cpu, dt, how)}


def main():
    args = parse()
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
    from fastbox_amd import CosmoBox, default_cosmo

    N = args.nsamp
    if args.mode == "slab":
        return slab_main(args, rank, world, local_rank, torch, dist, np)
    from fastbox_amd.device import new_stream
    boxes = [CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, redshift=0., realise_now=False,
                      precision=args.precision, rng="device", seed=1000 * (rank + 1) + i, device=local_rank,
                      stream=(new_stream() if args.streams > 1 else None)) for i in range(max(1, args.streams))]
    eng = boxes[0].engine
    counter = [0]

    def step():
        box = boxes[counter[0] % len(boxes)]
        counter[0] += 1
        dx = box.realise_density()
        return box.binned_power_spectrum(delta_x=box.lognormal(dx), nbins=args.nbins, wait=False)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(args.warmup, len(boxes))):
        step().result()
    counter[0] = 0
    fence()
    ev_stride = 1 if args.all_kernel_events else max(1, args.kernel_event_stride)
    eng.profile_start(None if args.all_kernel_events else ["fft_strided"], stride=ev_stride)
    t0 = time.perf_counter()
    acc = np.zeros(args.nbins - 1)
    pending = [step() for _ in range(args.steps)]
    for pnd in pending:
        kc, pk, err = pnd.result()
        acc += np.nan_to_num(pk)
    prof = eng.profile_stop()                    # synchronises the launch stream
    plain_launches = eng.profile_seen() if ev_stride > 1 else prof["fft_strided"][1]   # all of them, bracketed or not
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=args._reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        # roofline of the dominant kernel class: strided FFT pass over a half spectrum reads and
        # writes N*N*(N/2+1) complex values once each (DESIGN.md "Algorithmic bytes")
        s = 4 if args.precision == "f32" else 8
        ms, launches = prof["fft_strided"]
        # a step holds two such passes (inverse and forward y); each is launched once per x-plane batch
        # (fb_fft_launch.inc yz_passes), so one launch moves its share of the two passes' bytes
        alg_bytes = 2 * (2.0 * N * N * (N // 2 + 1) * 2 * s) * args.steps / max(plain_launches, 1)
        achieved = alg_bytes / (ms / max(launches, 1) * 1e-3) / 1e9 if ms > 0 else None
        total_ms = sum(v[0] for v in prof.values()) * (plain_launches / max(launches, 1) if ev_stride > 1 else 1.)
        # HBM traffic of the same kernel from the PMC counters: collected off-line in two separate
        # rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) over this very command and committed under
        # profiles/; FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM).
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01g_pmc_fetch_write_summary.json")))
            key = [k for k in pmc if "k_fft_strided<float, %d, 0" % N in k]
            if key and args.precision == "f32":
                c = pmc[key[0]]
                traffic = (2.0 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024.0
        except Exception:
            traffic = None
        # the committed rocprofv3 --kernel-trace --stats summary of this command, for the cross-check the two
        # timings owe each other (the event bracket also holds the ~3 us between the event and the kernel's start)
        rocprof_us = None
        try:
            import csv
            with open(os.path.join(ROOT, "profiles", "r01g_kernel_stats.csv")) as fh:
                for row in csv.DictReader(fh):
                    if ("k_fft_strided<%s, %d, 0" % ("float" if s == 4 else "double", N)) in row["Name"]:
                        rocprof_us = float(row["AverageNs"]) * 1e-3
        except Exception:
            rocprof_us = None
        line = {
            "metric": "%d^3 box realisations/sec (gen + log-normal + P(k))" % N,
            "value": world * args.steps / dt, "unit": "boxes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%d^3 Gaussian box (device Threefry noise, stand-in EH P(k), L=1000 Mpc) + "
                                   "log-normal transform + binned P(k), nbins=%d" % (N, args.nbins),
                       "nsamp": N, "parallelism": "replicas x%d" % world, "streams_per_gpu": len(boxes)},
            "roofline": {"bound": "hbm", "kernel": "k_fft_strided (x/y FFT pass, half spectrum)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": traffic,
                         "algorithmic_bytes": alg_bytes,
                         "avg_launch_us": 1e3 * ms / max(launches, 1), "launches": plain_launches,
                         "launches_timed": launches, "rocprof_avg_launch_us": rocprof_us},
            # (sampled brackets: scaled from the timed launches to all of them)
            "kernel_ms_per_step": {k: round(v[0] / args.steps * (plain_launches / max(v[1], 1) if ev_stride > 1 else 1.), 4)
                                   for k, v in prof.items() if v[1]},
            "kernel_ms_total_per_step": round(total_ms / args.steps, 4),
        }
        # whole step against the HBM roofline: SURVEY 8(d)'s byte model for this workload is 5.0 sweeps of
        # N^3 complex values (this implementation moves 4.5: the z passes of realisation and estimate are one)
        sweep = float(N) ** 3 * 2 * s
        boxes_per_s = line["value"] / world            # per GPU
        line["pipeline_roofline"] = {"bound": "hbm", "model_sweeps": 5.0, "moved_sweeps": 4.5,
                                     "model_bytes_per_box": 5.0 * sweep,
                                     "achieved": 5.0 * sweep * boxes_per_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": 5.0 * sweep * boxes_per_s / 1e9 / HBM_PEAK_GBS}
        if not args.no_cpu_baseline and world == 1:       # rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(N, args.cpu_nsamp or (N if N <= 512 else 256), args.nbins)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def slab_main(args, rank, world, local_rank, torch, dist, np):
    """One box over all ranks: SlabBox (x-slabs <-> k_y-slabs, one all-to-all per transform)."""
    from fastbox_amd import default_cosmo
    from fastbox_amd.distributed import SlabBox
    N = args.nsamp
    torch.cuda.set_device(local_rank)
    box = SlabBox(default_cosmo, box_scale=1e3, nsamp=N, precision=args.precision, seed=1000, rank=rank, world=world,
                  device=local_rank)

    def step():
        return box.realise_and_power(nbins=args.nbins, lognormal=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kc, pk, err = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=args._reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        s = 4 if args.precision == "f32" else 8
        print(json.dumps({
            "metric": "%d^3 box realisations/sec (gen + log-normal + P(k)), one box over all GPUs" % N,
            "value": args.steps / dt, "unit": "boxes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%d^3 Gaussian box + log-normal + binned P(k), slab-decomposed FFT, one all-to-all "
                                   "per transform, P(k) synchronised every step" % N, "nsamp": N,
                       "parallelism": "slab x%d" % world,
                       "all_to_all_bytes_per_rank_pair": (N // world) ** 2 * ((N // 2 + 16) // 16 * 16) * 2 * s},
            "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
