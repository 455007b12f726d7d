"""Diagnostic (GPU, library built with -DFB_STAMPS): where does a plain strided pass spend its
time?  Prints per-phase durations (s_memtime ticks = 100 MHz ref clock -> 10 ns) per workgroup."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, _lib
from fastbox_amd.device import HALF

N = 512
axis = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
box = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f32", rng="device")
eng = box.engine
h = eng.empty(HALF)
dx = box.realise_density()
box.binned_power_spectrum(delta_x=dx)            # sets bins/thresholds
for _ in range(3):
    _lib.call("fb_debug_strided_pass", eng._plan, h.ptr, axis, mode, eng.stream)
nt = 17 * N
if mode == 2:
    raw = np.zeros(2 * 20 * nt + nt * 32, dtype=np.int64)
    _lib.call("fb_debug_read_stamps", eng._plan, raw.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), raw.size)
    full = raw[2 * 20 * nt:].reshape(nt, 32)
    st = full[:, :8]
    # per wave: end of its binning relative to the barrier that opened the binning phase (stamp 5)
    wv = (full[:, 8:24] - full[:, 5:6]).astype(float)
    print("binning phase per wave (ticks after the staging barrier): wave  median  mean  p90  max")
    for w in range(16):
        print("   wave %2d %8.0f %8.0f %8.0f %8.0f" % (w, np.median(wv[:, w]), wv[:, w].mean(), np.percentile(wv[:, w], 90), wv[:, w].max()))
    slow = wv.max(axis=1)
    print("slowest wave of a workgroup: median %.0f mean %.0f p90 %.0f; which wave is slowest (histogram):" % (np.median(slow), slow.mean(), np.percentile(slow, 90)),
          np.bincount(wv.argmax(axis=1), minlength=16))
else:
    full = np.zeros((nt, 32), dtype=np.int64)
    _lib.call("fb_debug_read_stamps", eng._plan, full.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), full.size)
    st = full[:, :8]
t0 = st[:, 0].min()
names = ["start->loads issued", "loads issued->landed", "landed->tables", "fft stages",
         "stores issued | p staged", "stores drained | binned", "(bin) partials written"]
d = np.diff(st[:, :8], axis=1).astype(float)
tick = 1.0  # report raw ticks and convert assuming 100 MHz
print("workgroups", nt, "kernel span ticks", st[:, 6].max() - t0)
for i, n in enumerate(names):
    print("%-24s median %8.0f  mean %8.0f  p90 %8.0f ticks" % (n, np.median(d[:, i]), d[:, i].mean(), np.percentile(d[:, i], 90)))
life = (st[:, 7 if mode == 2 else 6] - st[:, 0]).astype(float)
bx = np.arange(nt) % 17
for k in (0, 1, 8, 16):
    sel = bx == k
    print("tiles with bx=%2d: median lifetime %8.0f, binning phase %8.0f" % (k, np.median(life[sel]), np.median(d[sel, 5])))
print("workgroup lifetime       median %8.0f  mean %8.0f ticks" % (np.median(life), life.mean()))
starts = np.sort(st[:, 0] - t0)
print("start times: first 5", starts[:5], " 512th", starts[511], " 513th", starts[512], " last", starts[-1])
