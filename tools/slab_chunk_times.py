"""Tuning aid (one GPU): what ONE rank of a P-rank slab job computes per k_z chunk, timed with the launches a real rank
would make (P virtual ranks in this process; rank 0's passes bracketed by events, the exchange done by block copies).
Feeds the strong-scaling projection of DESIGN.md section 6.   python tools/slab_chunk_times.py [N] [P] [C] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fastbox_amd import default_cosmo
from fastbox_amd.distributed import HipSlabOps, SlabBox, run_virtual_chunked

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
C = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
nb = 20
boxes = [SlabBox(default_cosmo, box_scale=1e3, nsamp=N, precision="f32", seed=3, rank=r, world=P,
                 ops_factory=lambda g, PP, rr: HipSlabOps(g, PP, rr, precision="f32", device=0), chunks=C) for r in range(P)]
for b in boxes:
    b._pk_setup(nb, None)
run_virtual_chunked(boxes, nb, True)           # fills every buffer with a real realisation (and warms the kernels up)
b = boxes[0]
A, B = b._chunk_views(0), b._chunk_views(1)
tab = b._chunk_tab
res = b.ops.new_results(2 * nb + 1)
real = b.ops.new_real()


def timed(fn):
    ts = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:]))


rows = []
for c, (t0, nt) in enumerate(tab):
    g = timed(lambda: b.ops.x_generate_chunk(A[c], 3, 0, t0, nt))
    yi = timed(lambda: b.ops.y_inverse_chunk(B[c], b._half, t0, nt))
    yf = timed(lambda: b.ops.y_forward_chunk(b._half, A[c], t0, nt))
    xb = timed(lambda: b.ops.x_bin_chunk(B[c], t0, nt, True, True, res))
    nbytes = A[c].numel() * A[c].element_size()
    rows.append((c, nt, g, yi, yf, xb, nbytes))
z = timed(lambda: b.ops.z_pass(b._half, real, 2, True, res[2 * nb:]))
print("N = %d, P = %d ranks, C = %d chunks; one rank's passes, ms (median of %d)" % (N, P, C, reps))
print("chunk tiles   GEN-x   y-inv   y-fwd   BIN-x   all-to-all bytes per rank (sent = received), MB")
for c, nt, g, yi, yf, xb, nbytes in rows:
    print("%5d %5d %7.3f %7.3f %7.3f %7.3f   %8.1f" % (c, nt, g, yi, yf, xb, nbytes * (P - 1) / P / 1e6))
tg, ty, tf, tx = (sum(r[k] for r in rows) for k in (2, 3, 4, 5))
tot_bytes = sum(r[6] for r in rows) * (P - 1) / P
print("sums  GEN-x %.3f  y-inv %.3f  z (c2r + r2c, exp) %.3f  y-fwd %.3f  BIN-x %.3f  = %.3f ms of compute per rank and step"
      % (tg, ty, z, tf, tx, tg + ty + z + tf + tx))
for eff in (1.0, 0.6):
    bw = 153e9 * eff * (P - 1 if P <= 8 else 7)            # one xGMI link per peer, each direction
    ta = tot_bytes / bw * 1e3
    # per exchange: the first chunk's transfer can start after its own pass, the last chunk's pass must wait for its
    # transfer: exposed = the part of the transfers the passes beside them do not cover
    per_chunk = [r[6] * (P - 1) / P / bw * 1e3 for r in rows]
    inv = rows[0][2] + sum(max(per_chunk[c], (rows[c + 1][2] if c + 1 < len(rows) else 0.0)) for c in range(len(rows))) \
        if len(rows) else 0.0
    print("links at %.0f %% of 153 GB/s: one all-to-all %.3f ms per rank (%.1f MB); both exchanges un-overlapped %.3f ms; "
          "inverse side overlapped (GEN-x(c+1) beside all-to-all(c)) ~%.3f ms against %.3f + %.3f in sequence"
          % (100 * eff, ta, tot_bytes / 1e6, 2 * ta, inv, tg, ta))
