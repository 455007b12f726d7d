"""BASELINE config 5 on the GPUs at hand: R realisations of an N^3 Gaussian box, P(k) of each, mean and
covariance of the band powers (Welford, fp64 on the host).  One process per GPU; ranks draw disjoint seeds and
the moment sums are combined at the end (no data-path collective).

    python tools/montecarlo_cov.py --nsamp 2048 --realisations 1000
    python -m torch.distributed.run --nproc-per-node 8 tools/montecarlo_cov.py --nsamp 2048 --realisations 1000
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsamp", type=int, default=512)
    ap.add_argument("--realisations", type=int, default=200)
    ap.add_argument("--nbins", type=int, default=20)
    ap.add_argument("--batch", type=int, default=50, help="spectra queued before their bin sums are fetched")
    ap.add_argument("--lognormal", action="store_true")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    import torch
    backend = os.environ.get("FASTBOX_BENCH_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")   # as bench.py
    if os.environ.get("FASTBOX_BENCH_ONE_DEVICE"):
        local = 0
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend)
    from fastbox_amd import CosmoBox, default_cosmo
    box = CosmoBox(default_cosmo, box_scale=1e3, nsamp=args.nsamp, realise_now=False, rng="device",
                   seed=7000 + rank, device=local)
    mine = [r for r in range(args.realisations) if r % world == rank]
    n, mean, m2, kc = 0, None, None, None
    box.binned_power_spectrum(delta_x=box.realise_density(), nbins=args.nbins)          # warm-up
    box.engine.sync()
    t0 = time.perf_counter()
    for start in range(0, len(mine), args.batch):
        pend = []
        for _ in mine[start:start + args.batch]:
            dx = box.realise_density()
            pend.append(box.binned_power_spectrum(delta_x=box.lognormal(dx) if args.lognormal else dx,
                                                  nbins=args.nbins, wait=False))
        for p in pend:
            kc, pk, _ = p.result()
            x = np.nan_to_num(pk)
            if mean is None:
                mean, m2 = np.zeros_like(x), np.zeros((x.size, x.size))
            n += 1
            d = x - mean
            mean += d / n
            m2 += np.outer(d, x - mean)
    dt = time.perf_counter() - t0
    if world > 1:          # combine (n, mean, M2) of the ranks: Chan et al. pairwise update via raw sums
        s = torch.tensor(np.concatenate([[n, dt], n * mean, (m2 + n * np.outer(mean, mean)).ravel()]))
        if backend == "nccl":
            s = s.cuda(local)
        dist.all_reduce(s)
        tmax = torch.tensor([dt], dtype=torch.float64, device=s.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())                                       # the slowest rank
        s = s.cpu().numpy()
        nt = s[0]
        mean = s[2:2 + mean.size] / nt
        m2 = s[2 + mean.size:].reshape(m2.shape) - nt * np.outer(mean, mean)
        n = int(nt)
    if rank == 0:
        cov = m2 / (n - 1)
        sig = np.sqrt(np.diag(cov))
        ok = sig > 0
        corr = cov[np.ix_(ok, ok)] / np.outer(sig[ok], sig[ok])
        off = corr[~np.eye(corr.shape[0], dtype=bool)]
        print("N=%d  %d realisations on %d GPU(s): %.2f s  (%.2f boxes/s)" % (args.nsamp, n, world, dt, n / dt))
        print("k centres        :", np.array2string(kc[ok][:6], precision=4), "...")
        print("mean P(k)        :", np.array2string(mean[ok][:6], precision=4), "...")
        print("sigma/P          :", np.array2string((sig[ok] / mean[ok])[:6], precision=3), "...")
        print("off-diagonal correlation coefficients: mean %.4f, rms %.4f (Gaussian field: 0 +- 1/sqrt(R) = %.4f)"
              % (off.mean(), np.sqrt((off ** 2).mean()), 1 / np.sqrt(n)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
