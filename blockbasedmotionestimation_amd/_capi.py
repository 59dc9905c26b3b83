"""ctypes binding of the C-ABI in include/bbme.h (libbbme.so, built in-tree by build.py).

There is no fallback of any kind: if the shared library is missing the import fails, and
every compute entry point returns BBME_ERR_HIP when no gfx950 device is usable.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BBME_LIB") or os.path.join(_PKG, "libbbme.so")     # BBME_LIB: development builds
MAX_LEVELS = 8

OK, ERR_INVALID, ERR_PADDING, ERR_ODD_PADDING, ERR_DEGENERATE, ERR_HIP, ERR_IO, ERR_STATE, ERR_UNSUPPORTED = \
    0, -1, -2, -3, -4, -5, -6, -7, -8


class BbmeError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("bbme status %d: %s" % (status, message))
        self.status = status
        self.message = message


class Params(C.Structure):
    _fields_ = [("num_levels", C.c_int),
                ("block_size", C.c_int * MAX_LEVELS),
                ("search_size", C.c_int * MAX_LEVELS)]


def make_params(search_size, block_size):
    if len(search_size) != len(block_size):
        raise ValueError("search_size and block_size must have one entry per level")
    p = Params()
    p.num_levels = len(block_size)
    if p.num_levels > MAX_LEVELS:
        raise ValueError("at most %d levels" % MAX_LEVELS)
    for i, (s, b) in enumerate(zip(search_size, block_size)):
        p.search_size[i] = int(s)
        p.block_size[i] = int(b)
    return p


# every symbol include/bbme.h declares: name -> (restype, argtypes)
_P = C.POINTER
_ctx = C.c_void_p
SIGNATURES = {
    "bbme_version": (C.c_char_p, []),
    "bbme_last_error": (C.c_char_p, []),
    "bbme_plan_padding": (C.c_int, [C.c_int, C.c_int, _P(Params)] + [_P(C.c_int)] * 4),
    "bbme_pad_zero_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bbme_pyr_down_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bbme_resize_x4_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bbme_flo_read": (C.c_int, [C.c_char_p, _P(C.c_int), _P(C.c_int), _P(_P(C.c_float))]),
    "bbme_flo_write": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    "bbme_flo_writer_create": (C.c_int, [_P(C.c_void_p)]),
    "bbme_flo_writer_create_pool": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "bbme_flo_writer_ticket": (C.c_int, [C.c_void_p, _P(C.c_ulonglong)]),
    "bbme_flo_writer_wait_ticket": (C.c_int, [C.c_void_p, C.c_ulonglong]),
    "bbme_flo_writer_submit": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "bbme_flo_writer_submit_cells": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                               C.c_int, C.c_int]),
    "bbme_flo_writer_wait": (C.c_int, [C.c_void_p]),
    "bbme_flo_writer_destroy": (C.c_int, [C.c_void_p]),
    "bbme_calculate_mse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _P(C.c_double)]),
    "bbme_motion_to_color": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, _P(C.c_float)]),
    "bbme_ppm_write_bgr": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    "bbme_subsample_div4": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_int]),
    "bbme_free": (None, [C.c_void_p]),
    "bbme_spiral_host": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, _P(C.c_int)]),
    "bbme_search_plan_host": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, _P(C.c_int), C.c_void_p, _P(C.c_int), _P(C.c_int)]),
    "bbme_search_plan_host_waves": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, _P(C.c_int), C.c_void_p, _P(C.c_int),
                                              _P(C.c_int)]),
    "bbme_create": (C.c_int, [_P(Params), C.c_int, C.c_int, C.c_int, _P(_ctx)]),
    "bbme_create_batch": (C.c_int, [_P(Params), C.c_int, C.c_int, C.c_int, C.c_int, _P(_ctx)]),
    "bbme_batch_size": (C.c_int, [_ctx, _P(C.c_int)]),
    "bbme_destroy": (C.c_int, [_ctx]),
    "bbme_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "bbme_get_stream": (C.c_int, [_ctx, _P(C.c_void_p)]),
    "bbme_get_geometry": (C.c_int, [_ctx] + [_P(C.c_int)] * 4),
    "bbme_level_geometry": (C.c_int, [_ctx, C.c_int] + [_P(C.c_int)] * 4),
    "bbme_set_frames_host": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int]),
    "bbme_set_frames_device": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int]),
    "bbme_set_frames_host_pair": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "bbme_set_frames_host_async": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "bbme_set_frames_device_pair": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "bbme_flow_device_pair": (C.c_int, [_ctx, C.c_int, _P(C.c_void_p)]),
    "bbme_get_flow_host_pair": (C.c_int, [_ctx, C.c_int, C.c_void_p]),
    "bbme_cells_device_pair": (C.c_int, [_ctx, C.c_int, _P(C.c_void_p)]),
    "bbme_get_cells_host_pair": (C.c_int, [_ctx, C.c_int, C.c_void_p]),
    "bbme_level_planes_device": (C.c_int, [_ctx, C.c_int, _P(C.c_void_p), _P(C.c_void_p)]),
    "bbme_set_level_planes_host": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p]),
    "bbme_get_level_planes_host": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p]),
    "bbme_estimate": (C.c_int, [_ctx]),
    "bbme_synchronize": (C.c_int, [_ctx]),
    "bbme_flow_device": (C.c_int, [_ctx, _P(C.c_void_p)]),
    "bbme_get_flow_host": (C.c_int, [_ctx, C.c_void_p]),
    "bbme_get_cells_host": (C.c_int, [_ctx, C.c_void_p]),
    "bbme_calculate_mse_device": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, _P(C.c_double)]),
    "bbme_cells_device": (C.c_int, [_ctx, _P(C.c_void_p)]),
    "bbme_expand_cells_device": (C.c_int, [_ctx, C.c_void_p, C.c_void_p]),
    "bbme_expand_cells_device_on": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bbme_stage_search": (C.c_int, [_ctx, C.c_int]),
    "bbme_stage_regularize": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int]),
    "bbme_stage_get_mvs": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p]),
    "bbme_stage_set_mvs": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p]),
    "bbme_stage_expand": (C.c_int, [_ctx]),
    "bbme_last_sweep_passes": (C.c_int, [_ctx, _P(C.c_int)]),
    "bbme_sweep_stats": (C.c_int, [_ctx, _P(C.c_uint)]),
    "bbme_set_profiling": (C.c_int, [_ctx, C.c_int]),
    "bbme_get_timings": (C.c_int, [_ctx] + [_P(C.c_float)] * 5),
    "bbme_probe_rates": (C.c_int, [C.c_int, _P(C.c_double)]),
    "bbme_probe_search_loops": (C.c_int, [C.c_int, _P(C.c_double)]),
    "bbme_probe_latency": (C.c_int, [C.c_int, _P(C.c_ulonglong)]),
    "bbme_probe_xcd": (C.c_int, [C.c_int, _P(C.c_int), _P(C.c_int)]),
    "bbme_set_search_mode": (C.c_int, [_ctx, C.c_int]),
    "bbme_set_regularizer_mode": (C.c_int, [_ctx, C.c_int]),
    "bbme_set_speculation": (C.c_int, [_ctx, C.c_int]),
    "bbme_set_relaxation": (C.c_int, [_ctx, C.c_int]),
    "bbme_wait_for_stream": (C.c_int, [_ctx, C.c_void_p]),
    "bbme_calibrate_read": (C.c_int, [C.c_int, C.c_uint, C.c_int]),
    "bbme_selftest_isa": (C.c_int, [C.c_int, _P(C.c_int)]),
}

_lib = None


def lib():
    """Load libbbme.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -m blockbasedmotionestimation_amd.build` "
                              "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
        # torch ships its own libamdhip64.so.7 / libhsa-runtime64.so.  Two HIP runtimes in one
        # process cannot both own the GPU, so torch's is loaded first and libbbme.so's
        # DT_NEEDED libamdhip64.so.7 then resolves to that same runtime (same SONAME).
        # torch is plumbing here (device memory, streams, torch.distributed), never compute.
        if os.environ.get("BBME_NO_TORCH_PRELOAD", "0") == "0":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise BbmeError(status, lib().bbme_last_error().decode("utf-8", "replace"))
    return status
