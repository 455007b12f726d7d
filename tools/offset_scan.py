"""Tuning aid (GPU): does the time of the fused z pass (reads the half spectrum, writes delta_x and the half spectrum)
and of the y pass depend on WHERE the two buffers lie relative to each other?  One arena, the real buffer at a sweep
of offsets behind the half-spectrum buffer.  python tools/offset_scan.py [N]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastbox_amd import CosmoBox, default_cosmo, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
box = CosmoBox(cosmo=default_cosmo, box_scale=1e3, nsamp=N, realise_now=False, precision="f32", rng="device", seed=1)
eng = box.engine
box.binned_power_spectrum(delta_x=box.lognormal(box.realise_density()), nbins=20)      # tables, bins
hb, rb = eng.nbytes["half"], eng.nbytes["real"]
arena = ctypes.c_void_p()
span = hb + rb + (512 << 20)
_lib.call("fb_malloc", ctypes.byref(arena), span)
res = ctypes.c_void_p()
_lib.call("fb_malloc", ctypes.byref(res), 4096)
base = arena.value
print("arena at 0x%x, half %d B, real %d B" % (base, hb, rb))
align = lambda x, a: (x + a - 1) // a * a
for label, off in (("packed (256 B aligned)", align(hb, 256)), ("+4 KiB", align(hb, 4096) + 4096), ("2 MiB aligned", align(hb, 2 << 20)),
                   ("2 MiB + 128 B", align(hb, 2 << 20) + 128), ("2 MiB + 1 KiB", align(hb, 2 << 20) + 1024),
                   ("2 MiB + 4 KiB", align(hb, 2 << 20) + 4096), ("2 MiB + 16 KiB", align(hb, 2 << 20) + 16384),
                   ("2 MiB + 64 KiB", align(hb, 2 << 20) + 65536), ("2 MiB + 256 KiB", align(hb, 2 << 20) + (256 << 10)),
                   ("2 MiB + 1 MiB", align(hb, 2 << 20) + (1 << 20)), ("64 MiB aligned", align(hb, 64 << 20)),
                   ("64 MiB + 8 MiB", align(hb, 64 << 20) + (8 << 20)), ("256 MiB aligned", align(hb, 256 << 20))):
    if off + rb > span:
        continue
    half, real = base, base + off
    times = []
    for rep in range(3):
        eng.profile_start()
        for i in range(10):
            _lib.call("fb_realise_density_begin", eng._plan, 1, i, ctypes.c_void_p(half), eng.stream)
            _lib.call("fb_power_spectrum_pending", eng._plan, ctypes.c_void_p(half), ctypes.c_void_p(real), 1, res, eng.stream)
        prof = eng.profile_stop()
        times.append((prof["fft_contig"][0] / 10, prof["fft_strided"][0] / 20))
    t = min(times)
    print("real = half + %-22s z pass %6.1f us   y pass %6.1f us" % (label + ":", t[0] * 1e3, t[1] * 1e3))
