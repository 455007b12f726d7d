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
nt = 16 * N
full = np.zeros((nt, 32), dtype=np.int64)
_lib.call("fb_debug_read_stamps", eng._plan, full.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), full.size)
if os.environ.get("FB_STAMPS_DUMP"):
    np.save(os.environ["FB_STAMPS_DUMP"], full)
st = full[:, :8].astype(np.int64)
hw = full[:, 24]
print("XCC_ID register values seen:", np.unique(full[:, 25] & 0xf))
xcc = np.arange(nt) % 8                        # workgroups go to the XCDs round robin
for x in np.unique(xcc):                       # every XCD counts from its own origin: align on each XCD's first start
    st[xcc == x] -= st[xcc == x][:, 0].min()
print("start stamps after alignment: percentiles 0/1/50/99/100:", np.percentile(st[:, 0], [0, 1, 50, 99, 100]))
print("end   stamps after alignment: percentiles 0/1/50/99/100:", np.percentile(st[:, 6], [0, 1, 50, 99, 100]))
for x in range(8):
    m = xcc == x
    print("  XCD %d: register says %s, starts %d..%d, ends %d..%d" % (x, np.unique(full[m, 25] & 0xf), st[m, 0].min(), st[m, 0].max(), st[m, 6].min(), st[m, 6].max()))
span = int(np.percentile(st[:, 6], 99.5))
keep = (st[:, 6] <= span) & (st[:, 0] >= 0)
print("kept %d of %d workgroups" % (keep.sum(), len(keep)))
st, hw, xcc = st[keep], hw[keep], xcc[keep]
assert 0 < span < 10 ** 7, "time stamps out of range: %d" % span
cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)).astype(np.int64)
print("workgroups %d, span %d ticks (%.1f us), distinct CUs %d" % (len(st), span, span / 100.0, len(np.unique(cu))))
for nm, a, b in (("start->fft input ready", 0, 3), ("fft", 3, 4), ("epilogue, stores issued", 4, 5), ("stores drained", 5, 6), ("lifetime", 0, 6)):
    d = (st[:, b] - st[:, a]).astype(float)
    print("%-26s median %7.0f mean %7.0f p10 %7.0f p90 %7.0f ticks" % (nm, np.median(d), d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
ts = np.arange(0, span, 20)
act = ((st[:, 0][None, :] <= ts[:, None]) & (ts[:, None] < st[:, 6][None, :])).sum(1)
drain = ((st[:, 5][None, :] <= ts[:, None]) & (ts[:, None] < st[:, 6][None, :])).sum(1)
comp = ((st[:, 0][None, :] <= ts[:, None]) & (ts[:, None] < st[:, 5][None, :])).sum(1)
print("time(us)  resident  computing  draining")
for i in range(0, len(ts), max(1, len(ts) // 120)):
    print("%7.1f %9d %9d %9d" % (ts[i] / 100.0, act[i], comp[i], drain[i]))
mid = slice(len(ts) // 5, 4 * len(ts) // 5)
print("middle 60%% of the span: draining workgroups mean %.0f, std %.0f, min %d, max %d" % (drain[mid].mean(), drain[mid].std(), drain[mid].min(), drain[mid].max()))
# workgroups sharing a CU: start offsets of overlapping pairs
offs = []
for c in np.unique(cu):
    s = st[cu == c]
    s = s[np.argsort(s[:, 0])]
    for i in range(1, len(s)):
        if s[i, 0] < s[i - 1, 6]:
            offs.append((s[i, 0] - s[i - 1, 0]) / max(1.0, float(s[i - 1, 6] - s[i - 1, 0])))
offs = np.array(offs)
print("co-resident pairs %d: start offset / lifetime  median %.2f  p10 %.2f  p90 %.2f  (0 = lock step, 0.5 = alternating)"
      % (len(offs), np.median(offs), np.percentile(offs, 10), np.percentile(offs, 90)))
print("workgroups per CU: min %d max %d" % (np.bincount(np.unique(cu, return_inverse=True)[1]).min(), np.bincount(np.unique(cu, return_inverse=True)[1]).max()))
