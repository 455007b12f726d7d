"""BASELINE config 5 on the GPUs at hand: R realisations of an N^3 Gaussian box, P(k) of each, mean and
covariance of the band powers (fastbox_amd/montecarlo.py: Welford sums, checkpoint / resume).  One process per GPU.

    python tools/montecarlo_cov.py --nsamp 2048 --realisations 1000 --checkpoint /tmp/mc.npz
    python -m torch.distributed.run --nproc-per-node 8 tools/montecarlo_cov.py --nsamp 2048 --realisations 1000
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsamp", type=int, default=512)
    ap.add_argument("--realisations", type=int, default=200)
    ap.add_argument("--nbins", type=int, default=20)
    ap.add_argument("--batch", type=int, default=50, help="spectra queued before their bin sums are fetched")
    ap.add_argument("--lognormal", action="store_true")
    ap.add_argument("--checkpoint", default=None, help="state file (per rank: .rankN is appended); resumes if it exists")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    import torch
    backend = os.environ.get("FASTBOX_BENCH_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")   # as bench.py
    if os.environ.get("FASTBOX_BENCH_ONE_DEVICE"):
        local = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend)
    from fastbox_amd import CosmoBox, default_cosmo, montecarlo
    box = CosmoBox(default_cosmo, box_scale=1e3, nsamp=args.nsamp, realise_now=False, rng="device", seed=7000, device=local)
    box.binned_power_spectrum(delta_x=box.realise_density(), nbins=args.nbins)          # warm-up
    box.engine.sync()
    ck = None if args.checkpoint is None else "%s.rank%d" % (args.checkpoint, rank)
    acc, kc, dt = montecarlo.run(box, args.realisations, nbins=args.nbins, lognormal=args.lognormal, batch=args.batch,
                                 rank=rank, world=world, checkpoint=ck)
    if world > 1:
        acc = montecarlo.combine(acc, dist, device=("cuda:%d" % local) if backend == "nccl" else None)
        tmax = torch.tensor([dt], dtype=torch.float64, device=("cuda:%d" % local) if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())                                       # the slowest rank
    if rank == 0:
        n, mean = acc.n, acc.mean
        cov = acc.covariance()
        sig = np.sqrt(np.diag(cov))
        ok = sig > 0
        corr = cov[np.ix_(ok, ok)] / np.outer(sig[ok], sig[ok])
        off = corr[~np.eye(corr.shape[0], dtype=bool)]
        print("N=%d  %d realisations on %d GPU(s): %.2f s in this call" % (args.nsamp, n, world, dt))
        print("k centres        :", np.array2string(kc[ok][:6], precision=4), "...")
        print("mean P(k)        :", np.array2string(mean[ok][:6], precision=4), "...")
        print("sigma/P          :", np.array2string((sig[ok] / mean[ok])[:6], precision=3), "...")
        print("off-diagonal correlation coefficients: mean %.4f, rms %.4f (Gaussian field: 0 +- 1/sqrt(R) = %.4f)"
              % (off.mean(), np.sqrt((off ** 2).mean()), 1 / np.sqrt(n)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
