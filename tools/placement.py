"""Tuning aid (GPU): does the time of one strided pass depend on where the buffer lies?
Allocates several half-spectrum buffers (all kept alive, so every one has a different address) and
times the same in-place y pass and x pass on each.  python tools/placement.py [N] [count]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastbox_amd import CosmoBox, default_cosmo, _lib
from fastbox_amd.device import HALF

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
box = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f32", rng="device")
eng = box.engine
keep = []
for i in range(count):
    h = eng.empty(HALF)
    keep.append(h)
    out = []
    for axis in (1, 0):
        for rep in range(2):
            eng.profile_start()
            for _ in range(10):
                _lib.call("fb_debug_strided_pass", eng._plan, h.ptr, axis, 0, eng.stream)
            prof = eng.profile_stop()
        out.append(sum(v[0] for v in prof.values()) / 10 * 1e3)
    print("buffer %2d at 0x%x (mod 2 MiB: 0x%06x)   y pass %6.1f us   x pass %6.1f us" %
          (i, h.ptr, h.ptr % (2 << 20), out[0], out[1]))
