"""Diagnostic (GPU, library built with -DFB_STAMPS): where does a plain strided pass spend its
time?  Prints per-phase durations (s_memtime ticks = 100 MHz ref clock -> 10 ns) per workgroup."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, _lib
from fastbox_amd.device import HALF

N = 512
axis = int(sys.argv[1]) if len(sys.argv) > 1 else 1
box = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f32", rng="device")
eng = box.engine
h = eng.empty(HALF)
for _ in range(3):
    _lib.call("fb_debug_strided_pass", eng._plan, h.ptr, axis, 0, eng.stream)
nt = 17 * N
st = np.zeros((nt, 8), dtype=np.int64)
_lib.call("fb_debug_read_stamps", eng._plan, st.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), st.size)
t0 = st[:, 0].min()
names = ["start->loads issued", "loads issued->landed", "landed->barrier", "fft stages", "stores issued", "stores->drained"]
d = np.diff(st[:, :7], axis=1).astype(float)
tick = 1.0  # report raw ticks and convert assuming 100 MHz
print("workgroups", nt, "kernel span ticks", st[:, 6].max() - t0)
for i, n in enumerate(names):
    print("%-24s median %8.0f  mean %8.0f  p90 %8.0f ticks" % (n, np.median(d[:, i]), d[:, i].mean(), np.percentile(d[:, i], 90)))
life = (st[:, 6] - st[:, 0]).astype(float)
print("workgroup lifetime       median %8.0f  mean %8.0f ticks" % (np.median(life), life.mean()))
starts = np.sort(st[:, 0] - t0)
print("start times: first 5", starts[:5], " 512th", starts[511], " 513th", starts[512], " last", starts[-1])
