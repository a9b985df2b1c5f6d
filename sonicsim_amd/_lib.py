"""ctypes binding of libsonicsim_hip.so (include/sonicsim_hip.h).

Importing this module does NOT initialise HIP (``SonicSet.py:154`` uses the 'spawn' start method;
worker processes re-import modules).  The library itself initialises lazily on the first call.
There is no CPU fallback: a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libsonicsim_hip.so")     # the product reads no environment switch; A/B builds: use_library()

FLAG_DEVICE_PTR = 0x1
FLAG_PATH_OS = 0x10
FLAG_PATH_DIRECT = 0x20
FLAG_GEOM_2048 = 0x40
FLAG_GEOM_4096 = 0x80
FLAG_LAYOUT_TC = 0x100
FLAG_GEOM_13 = 0x200
FLAG_GEOM_ASM = 0x400
FLAG_ASYNC_PLAN = 0x800
FLAG_META_DEVICE = 0x1000
FLAG_RESULT_DEVICE = 0x2000
FLAG_KEEP_SPEAKERS = 0x4000
FLAG_BANK_DEVICE = 0x8000
FLAG_ROW_SPECTRA = 0x10000
FLAG_NO_ROW_SPECTRA = 0x20000
FLAG_BACKGROUND = 0x40000

SS_EINVAL, SS_EHIP, SS_ENOMEM, SS_ENODEV = -1, -2, -3, -4

c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)


class SsRirParams(ctypes.Structure):
    _fields_ = [("P", ctypes.c_int32), ("C", ctypes.c_int32), ("L", ctypes.c_int32),
                ("fs", ctypes.c_float), ("rt60", ctypes.c_float), ("tail_gain", ctypes.c_float),
                ("rho", ctypes.c_float), ("seed", ctypes.c_uint32),
                ("delay", c_i32p), ("dgain", c_f32p)]


_SIGS = {
    "ss_version": (ctypes.c_int, []),
    "ss_last_error": (ctypes.c_char_p, []),
    "ss_init": (ctypes.c_int, [ctypes.c_int]),
    "ss_shutdown": (ctypes.c_int, []),
    "ss_convolve_moving_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                              ctypes.c_void_p]),
    "ss_convolve_moving_checked_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                      ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                                      ctypes.c_void_p, c_i64p]),
    "ss_set_task_queue": (ctypes.c_int, [ctypes.c_int]),
    "ss_workspace_lanes": (ctypes.c_int, [c_i32p, ctypes.c_int32]),
    "ss_stream_release": (ctypes.c_int, [ctypes.c_void_p]),
    "ss_set_host_pipe": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]),
    "ss_host_path_stats": (ctypes.c_int, [c_f64p, ctypes.c_int32]),
    "ss_host_alloc": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64]),
    "ss_host_free": (ctypes.c_int, [ctypes.c_void_p]),
    "ss_stream_open": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_uint32, ctypes.c_void_p]),
    "ss_stream_push": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_stream_info": (ctypes.c_int, [ctypes.c_void_p, c_i64p, ctypes.c_int32]),
    "ss_stream_close": (ctypes.c_int, [ctypes.c_void_p]),
    "ss_async_status": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ss_plan_status_last": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ss_convolve_moving_seg_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_convolve_fixed_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_rir_bank_synth_f32": (ctypes.c_int, [ctypes.POINTER(SsRirParams), ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_rir_bank_synth_peak_f32": (ctypes.c_int, [ctypes.POINTER(SsRirParams), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                                  ctypes.c_void_p]),
    "ss_rir_bank_synth_batch_f32": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(SsRirParams), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                                   ctypes.c_void_p]),
    "ss_peak_normalize_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_f32p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_divide_by_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_convolve_moving_seg_div_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                      ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                                      ctypes.c_void_p]),
    "ss_convolve_scene_f32": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_rms_db_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, c_f64p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_mix_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, c_f32p,
                                  ctypes.c_float, ctypes.c_void_p, c_f32p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_mean_channels_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_crop_rms_db_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, c_f64p,
                                          ctypes.c_uint32, ctypes.c_void_p]),
    "ss_mix_batch_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_int64, ctypes.c_int64, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_void_p, c_f32p,
                                        ctypes.c_uint32, ctypes.c_void_p]),
    "ss_rir_early_add_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_crop_sum_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_uint32, ctypes.c_void_p]),
    "ss_overlap_audio_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_resample_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f32p,
                                       c_i32p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_kweighted_block_power_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, c_f64p, c_i64p, c_i64p,
                                                    ctypes.c_int32, ctypes.c_double, c_f64p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_scale_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, c_f64p, ctypes.c_uint32,
                                    ctypes.c_void_p]),
    "ss_lufs_norm_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, c_f64p, c_i64p, c_i64p,
                                        ctypes.c_int32, ctypes.c_double, c_f64p, ctypes.c_double, c_f64p, ctypes.c_uint32,
                                        ctypes.c_void_p]),
    "ss_lufs_norm_batch_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, c_f64p,
                                              c_i64p, c_i64p, ctypes.c_int32, ctypes.c_double, c_f64p, c_f64p, ctypes.c_void_p, ctypes.c_uint32,
                                              ctypes.c_void_p]),
    "ss_lufs_norm_batch_sq_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, c_f64p,
                                                 c_i64p, c_i64p, ctypes.c_int32, ctypes.c_double, c_f64p, c_f64p, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_uint32, ctypes.c_void_p]),
    "ss_mix_presum_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, c_f32p, ctypes.c_float, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_lufs_norm_batch_sqx_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f64p,
                                                  c_i64p, c_i64p, ctypes.c_int32, ctypes.c_double, c_f64p, c_f64p, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_uint32, ctypes.c_void_p]),
    "ss_mix_onepass_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, c_f32p, ctypes.c_float, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "ss_gather_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "ss_gather_attach": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]),
    "ss_gather_slot": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]),
    "ss_gather_put": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "ss_gather_wait_src": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "ss_gather_flush": (ctypes.c_int, [ctypes.c_void_p]),
    "ss_gather_close": (ctypes.c_int, [ctypes.c_void_p]),
    "ss_prof_enable": (ctypes.c_int, [ctypes.c_int]),
    "ss_prof_read": (ctypes.c_int, [ctypes.c_int, c_i64p, c_f64p]),
    "ss_prof_seen": (ctypes.c_int, [ctypes.c_int, c_i64p]),
    "ss_prof_list": (ctypes.c_int, [ctypes.c_int, c_f64p, ctypes.c_int64, c_i64p]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def use_library(path: str):
    """Measurement tools only (tools/, bench.py --lib): load another build of the library -- the tuning build with its experiment
    switches, an A/B variant -- instead of the product's.  Must be called before the first load(); an explicit call, never an
    environment variable, so that nothing in a user's environment can swap the library under the drop-in modules."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_library() must be called before the library is first loaded")
    LIB_PATH = os.path.abspath(path)


def load():
    """Load the shared library (no HIP call is made).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m sonicsim_amd.build` "
            "(the MI355X renderer has no CPU fallback)")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  If THIS library were
    # loaded first it would pull in /opt/rocm's copies and torch would then run against a second HIP/HSA
    # runtime in the same process ("no ROCm-capable device").  Loading torch first makes the dynamic linker
    # resolve our NEEDED libamdhip64.so.7 to the already-loaded runtime, so torch tensors, streams and this
    # library share one runtime.  (Importing torch does not initialise HIP.)  Without torch installed the
    # system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().ss_last_error().decode("utf-8", "replace")


def check(rc: int):
    """Map C-ABI return codes to the exception types the reference would raise."""
    if rc == 0:
        return
    msg = last_error()
    if rc == SS_EINVAL:
        raise ValueError(msg)
    if rc == SS_ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(f"libsonicsim_hip error {rc}: {msg}")
