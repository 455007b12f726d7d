"""Diagnostic (GPU, library built with -DFB_STAMPS): are the workgroups of a strided pass in lock step?
Every workgroup stamps start / FFT done / stores issued / stores drained (s_memtime, 100 MHz) and the CU it ran on;
this prints how many workgroups sit in each phase over the kernel's span, and for each CU the phase offset of the
workgroups that shared it.

    python tools/phase_timeline.py <axis> <mode>      # axis 0 mode 1: the generator pass; axis 1 mode 0: a y pass
"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, _lib
from fastbox_amd.device import HALF

N = 512
axis = int(sys.argv[1]) if len(sys.argv) > 1 else 0
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
box = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f32", rng="device")
eng = box.engine
h = eng.empty(HALF)
dx = box.realise_density()
box.binned_power_spectrum(delta_x=dx)
for _ in range(3):
    _lib.call("fb_debug_strided_pass", eng._plan, h.ptr, axis, mode, eng.stream)
if mode == 2:                                  # the binning pass keeps its stamps behind the partial sums; unpacked layout
    nt = 17 * N
    raw = np.zeros(2 * 20 * nt + nt * 32, dtype=np.int64)
    _lib.call("fb_debug_read_stamps", eng._plan, raw.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), raw.size)
    full = raw[2 * 20 * nt:].reshape(nt, 32).copy()
    full[:, 6] = full[:, 7]                    # "end" = partial sums written
else:
    nt = 16 * N
    full = np.zeros((nt, 32), dtype=np.int64)
    _lib.call("fb_debug_read_stamps", eng._plan, full.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), full.size)
if os.environ.get("FB_STAMPS_DUMP"):
    np.save(os.environ["FB_STAMPS_DUMP"], full)
st = full[:, :8].astype(np.int64)
hw, xcc = full[:, 24], full[:, 25] & 0xf
cu = ((xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)).astype(np.int64)
ntx = 17 if mode == 2 else 16
bx = np.arange(nt) % ntx
life = (st[:, 6] - st[:, 0]).astype(float)
print("workgroups %d on %d CUs; s_memtime ticks are shader clocks (about 2.4 GHz), each CU counts from its own origin" % (nt, len(np.unique(cu))))
for nm, a, b in (("start->fft input ready", 0, 3), ("fft", 3, 4), ("epilogue", 4, 5), ("stores drained | binned", 5, 6), ("lifetime", 0, 6)):
    d = (st[:, b] - st[:, a]).astype(float)
    print("%-26s median %7.0f mean %7.0f p10 %7.0f p90 %7.0f ticks" % (nm, np.median(d), d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
print("lifetime by tile column (median):", " ".join("%d:%.0f" % (k, np.median(life[bx == k])) for k in range(ntx)))
spans, resid, one, gaps = {}, [], [], []
for c in np.unique(cu):
    s_ = st[cu == c]
    s_ = s_[np.argsort(s_[:, 0])]
    t0, t1 = s_[0, 0], s_[:, 6].max()
    spans[c] = t1 - t0
    resid.append((s_[:, 6] - s_[:, 0]).sum() / float(t1 - t0))
    ev = sorted([(a, 1) for a in s_[:, 0]] + [(b, -1) for b in s_[:, 6]])
    n, last, acc = 0, t0, [0, 0, 0, 0]
    for t, d in ev:
        acc[min(n, 3)] += t - last
        last = t
        n += d
    one.append((acc[0] + acc[1]) / float(t1 - t0))
    ends = np.sort(s_[:, 6])
    for i in range(2, len(s_)):                  # workgroup i takes the slot of the (i-2)th to finish
        gaps.append(s_[i, 0] - ends[i - 2])
sp = np.array(list(spans.values()), dtype=float)
print("span of a CU (first start to last end): min %.0f median %.0f max %.0f ticks; the kernel lasts as long as the slowest" % (sp.min(), np.median(sp), sp.max()))
for x in range(8):
    v = np.array([spans[c] for c in spans if (c >> 16) == x], dtype=float)
    print("   XCD %d: CUs %d, span median %.0f max %.0f" % (x, len(v), np.median(v), v.max()))
print("resident workgroups per CU: mean %.2f; fraction of the span with fewer than two: %.2f" % (np.mean(resid), np.mean(one)))
gaps = np.array(gaps, dtype=float)
print("refill gap (a workgroup ends -> the next starts on that CU): median %.0f p90 %.0f ticks" % (np.median(gaps), np.percentile(gaps, 90)))
