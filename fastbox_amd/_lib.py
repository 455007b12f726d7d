"""
ctypes binding of libfastbox_hip.so (C ABI declared in include/fastbox_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails the
binding raises.  The library is built in-tree by ``build_library()`` (hipcc,
--offload-arch=gfx950) into ``fastbox_amd/lib/``.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# FASTBOX_HIP_LIB: another build of the same sources (tuning variants, tools/build_variants.sh); default: the in-tree library
LIB_PATH = os.environ.get("FASTBOX_HIP_LIB") or os.path.join(_HERE, "lib", "libfastbox_hip.so")
CSRC = os.path.join(_HERE, "csrc")

c_void_p, c_int, c_double, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
c_u64, c_i64 = ctypes.c_uint64, ctypes.c_int64
P_double = ctypes.POINTER(ctypes.c_double)
P_i32 = ctypes.POINTER(ctypes.c_int32)

# name -> (restype, argtypes); must list every symbol of include/fastbox_hip.h
SIGNATURES = {
    "fb_version": (c_int, []),
    "fb_last_error": (ctypes.c_char_p, []),
    "fb_plan_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_double, c_double, c_double, c_int, c_int,
                               P_double, P_double, P_double, P_double]),
    "fb_plan_destroy": (c_int, [c_void_p]),
    "fb_half_pitch": (c_int, [c_void_p]),
    "fb_half_rows": (c_int, [c_void_p]),
    "fb_real_bytes": (c_i64, [c_void_p]),
    "fb_half_bytes": (c_i64, [c_void_p]),
    "fb_full_bytes": (c_i64, [c_void_p]),
    "fb_fft_c2c": (c_int, [c_void_p, c_void_p, c_int, c_double, c_void_p]),
    "fb_fft_r2c": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fb_fft_c2r": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_void_p]),
    "fb_set_amplitude_shells": (c_int, [c_void_p, P_double, c_i64]),
    "fb_set_amplitude_sym": (c_int, [c_void_p, P_double, c_i64]),
    "fb_set_amplitude_dense": (c_int, [c_void_p, c_void_p]),
    "fb_colour_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fb_colour_device": (c_int, [c_void_p, c_u64, c_u64, c_void_p, c_void_p]),
    "fb_set_bins": (c_int, [c_void_p, P_double, c_int, P_i32, P_i32, c_int]),
    "fb_bin_power": (c_int, [c_void_p, c_void_p, c_int, P_double, P_double, P_double, c_void_p]),
    "fb_bin_power_filtered": (c_int, [c_void_p, c_void_p, c_int, c_int, P_double, c_void_p, P_double, P_double,
                                      P_double, c_void_p]),
    "fb_apply_filter": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, P_double, c_void_p, c_void_p]),
    "fb_velocity_k": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_void_p]),
    "fb_potential_k": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fb_lognormal": (c_int, [c_void_p, c_void_p, c_void_p, P_double, c_void_p]),
    "fb_redshift_space": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_u64,
                                  c_int, c_void_p]),
    "fb_sum_real": (c_int, [c_void_p, c_void_p, c_int, P_double, c_void_p]),
    "fb_sumsq_half": (c_int, [c_void_p, c_void_p, P_double, c_void_p]),
    "fb_max_real": (c_int, [c_void_p, c_void_p, P_double, c_void_p]),
    "fb_expand_half": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fb_crop_full": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fb_realise_density_device": (c_int, [c_void_p, c_u64, c_u64, c_void_p, c_void_p, c_void_p]),
    "fb_realise_velocity_device": (c_int, [c_void_p, c_u64, c_u64, c_int, c_double, c_void_p, c_void_p, c_void_p]),
    "fb_realise_density_begin": (c_int, [c_void_p, c_u64, c_u64, c_void_p, c_void_p]),
    "fb_realise_density_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fb_power_spectrum_filtered": (c_int, [c_void_p, c_void_p, c_void_p, c_int, P_double, c_void_p, c_void_p, c_void_p]),
    "fb_power_spectrum_filtered_field": (c_int, [c_void_p, c_void_p, c_void_p, c_int, P_double, c_void_p, c_void_p, c_void_p]),
    "fb_fft_c2r_yz": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_void_p]),
    "fb_realise_velocity_begin": (c_int, [c_void_p, c_u64, c_u64, c_int, c_double, c_void_p, c_void_p]),
    "fb_power_spectrum_redshift_space": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double,
                                                 c_u64, c_int, c_int, P_double, c_void_p, c_int, c_void_p, c_void_p]),
    "fb_power_spectrum_pending": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "fb_montecarlo_power": (c_int, [c_void_p, c_u64, c_u64, c_u64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_i64, c_void_p]),
    "fb_power_spectrum_device": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "fb_bin_counts": (c_int, [c_void_p, P_double]),
    "fb_real_axpby": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_void_p]),
    "fb_real_multiply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fb_real_to_complex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fb_fft_transverse": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "fb_mask_transverse": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fb_beam_convolve": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fb_channel_means": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fb_channel_covariance": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fb_leading_eigenvectors": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fb_pca_clean": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "fb_sky_realise_map": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_u64, c_double, c_void_p, c_void_p, c_void_p]),
    "fb_sky_normal_map": (c_int, [c_void_p, c_void_p, c_u64, c_double, c_double, c_void_p, c_void_p]),
    "fb_sky_gaussian_filter": (c_int, [c_void_p, c_void_p, c_void_p, P_double, c_int, c_void_p]),
    "fb_sky_foreground_cube": (c_int, [c_void_p, c_void_p, c_void_p, c_double, P_double, c_void_p, c_void_p]),
    "fb_sky_noise_cube": (c_int, [c_void_p, P_double, c_void_p, c_u64, c_void_p, c_void_p]),
    "fb_slab_half_bytes": (c_i64, [c_void_p, c_int]),
    "fb_slab_kspace_bytes": (c_i64, [c_void_p, c_int]),
    "fb_slab_forward_local": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "fb_slab_inverse_local": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fb_slab_forward_packed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "fb_slab_inverse_packed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fb_slab_turnaround": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "fb_slab_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fb_slab_unpack": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fb_slab_x_pass": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fb_slab_x_generate": (c_int, [c_void_p, c_void_p, c_int, c_int, c_u64, c_u64, c_void_p]),
    "fb_slab_x_bin": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "fb_slab_tile_geometry": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "fb_slab_x_generate_chunk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_u64, c_u64, c_int, c_int, c_void_p]),
    "fb_slab_y_inverse_chunk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fb_slab_y_forward_chunk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fb_slab_z_pass": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fb_slab_x_bin_chunk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fb_set_plane_batching": (c_int, [c_void_p, c_int, c_int]),
    "fb_set_pass_schedule": (c_int, [c_void_p, c_int, c_int, c_int]),
    "fb_set_tile_rows": (c_int, [c_void_p, c_int]),
    "fb_comm_unique_id": (c_int, [c_void_p]),
    "fb_comm_create": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "fb_comm_destroy": (c_int, [c_void_p]),
    "fb_comm_info": (c_int, [c_void_p, P_i32, P_i32, P_i32]),
    "fb_slab_exchange_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, P_i32]),
    "fb_slab_exchange_wait": (c_int, [c_void_p, c_int, c_void_p]),
    "fb_slab_exchange": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "fb_allreduce_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fb_set_exp_shift": (c_int, [c_void_p, ctypes.c_double]),
    "fb_debug_strided_pass": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fb_debug_read_stamps": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_longlong), c_i64]),
    "fb_profile_select": (c_int, [c_void_p, ctypes.c_uint]),
    "fb_profile_sample": (c_int, [c_void_p, c_int, ctypes.POINTER(c_i64)]),
    "fb_profile_start": (c_int, [c_void_p]),
    "fb_profile_stop": (c_int, [c_void_p, c_void_p, P_double, ctypes.POINTER(c_i64), c_int]),
    "fb_malloc": (c_int, [ctypes.POINTER(c_void_p), c_size_t]),
    "fb_free": (c_int, [c_void_p]),
    "fb_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fb_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fb_memcpy_d2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fb_stream_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "fb_stream_create_priority": (c_int, [ctypes.POINTER(c_void_p), c_int]),
    "fb_stream_destroy": (c_int, [c_void_p]),
    "fb_stream_sync": (c_int, [c_void_p]),
    "fb_stream_wait_stream": (c_int, [c_void_p, c_void_p]),
    "fb_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "fb_device_get": (c_int, [ctypes.POINTER(c_int)]),
    "fb_device_set": (c_int, [c_int]),
}

FB_FILT_TABLE, FB_FILT_BEAM_HIGHPASS, FB_FILT_WEDGE, FB_FILT_TOPHAT = 0, 1, 2, 3


class FastBoxError(RuntimeError):
    """A libfastbox_hip call returned a non-zero status."""

    def __init__(self, fn, code, msg):
        RuntimeError.__init__(self, "%s failed (%d): %s" % (fn, code, msg))
        self.code = code


_lib = None


def build_library(verbose=False, jobs=8):
    """Compile every HIP source for gfx950 into fastbox_amd/lib/libfastbox_hip.so."""
    cmd = ["make", "-C", CSRC, "-j%d" % jobs]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if verbose or res.returncode:
        print(res.stdout)
    if res.returncode:
        raise RuntimeError("building libfastbox_hip.so failed")
    return LIB_PATH


def load():
    """dlopen the in-tree library and attach prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libfastbox_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` or `make -C fastbox_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7, and whichever copy is
    # loaded first serves both.  torch initialised on top of /opt/rocm's copy reports "No HIP GPUs",
    # so when torch is installed it goes first (it is needed anyway for the multi-GPU exchange).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(name, code):
    if code != 0:
        msg = load().fb_last_error()
        raise FastBoxError(name, code, msg.decode("utf-8", "replace") if msg else "")


def call(name, *args):
    check(name, getattr(load(), name)(*args))


class on_device(object):
    """``with on_device(d): ...`` -- the plan-less helpers (fb_malloc, fb_stream_create, fb_stream_wait_stream) act on
    the calling thread's current HIP device: make it `d` for the block and put the caller's back afterwards, so that
    torch / RCCL in the same thread keep the current device they had."""

    def __init__(self, device):
        self.device, self.prev = int(device), None

    def __enter__(self):
        prev = c_int(-1)
        call("fb_device_get", ctypes.byref(prev))
        if prev.value != self.device:
            call("fb_device_set", self.device)
            self.prev = prev.value
        return self

    def __exit__(self, *exc):
        if self.prev is not None and self.prev >= 0:
            call("fb_device_set", self.prev)
        return False


def device_count():
    n = c_int(0)
    code = load().fb_device_count(ctypes.byref(n))
    return n.value if code == 0 else 0
