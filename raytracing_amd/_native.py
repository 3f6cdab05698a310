"""ctypes binding of libraytracing_amd.so (the C ABI declared in include/rt_hip.h).

The library is the product: there is no Python or CPU implementation of the
kernel behind this module.  If the shared object has not been built the import
fails, and if no HIP device is present ``rt_create`` fails.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB points at another build of the same library (kernel experiments); default: in-tree
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "libraytracing_amd.so")


class RtError(RuntimeError):
    """A C-ABI call returned a non-zero status (the reference throws std::runtime_error)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"rt error {code}: {message}")
        self.code = code
        self.message = message


class Record(C.Structure):
    _fields_ = [("hit", C.c_int32), ("hash", C.c_uint32 * 3)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("sphere_tests", C.c_uint64), ("paths", C.c_uint64), ("list_rays", C.c_uint64)]


# name -> (restype, argtypes); every function include/rt_hip.h declares.
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_F = C.POINTER(C.c_float)
SIGNATURES = {
    "rt_abi_version": (C.c_int, []),
    "rt_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _PP]),
    "rt_destroy": (C.c_int, [_P]),
    "rt_resize": (C.c_int, [_P, C.c_int, C.c_int]),
    "rt_upload_spheres": (C.c_int, [_P, _P, C.c_int]),
    "rt_upload_materials": (C.c_int, [_P, _P, C.c_int]),
    "rt_set_uniforms": (C.c_int, [_P, _P]),
    "rt_set_max_depth": (C.c_int, [_P, C.c_int]),
    "rt_set_profile": (C.c_int, [_P, C.c_int]),
    "rt_set_kernel": (C.c_int, [_P, C.c_int]),
    "rt_set_option": (C.c_int, [_P, C.c_int, C.c_int]),
    "rt_set_tile_rows": (C.c_int, [_P, C.c_int, C.c_int]),
    "rt_set_frame_schedule": (C.c_int, [_P, C.c_int, C.c_int]),
    "rt_dispatch": (C.c_int, [_P]),
    "rt_dispatch_frames": (C.c_int, [_P, C.c_int]),
    "rt_sync": (C.c_int, [_P]),
    "rt_read_color": (C.c_int, [_P, _P]),
    "rt_read_accum": (C.c_int, [_P, _P]),
    "rt_write_accum": (C.c_int, [_P, _P]),
    "rt_device_buffers": (C.c_int, [_P, _PP, _PP]),
    "rt_bind_buffers": (C.c_int, [_P, _P, _P]),
    "rt_bind_buffers_next": (C.c_int, [_P, _P, _P]),
    "rt_sync_dispatch": (C.c_int, [_P, C.c_int]),
    "rt_wait_stream": (C.c_int, [_P, _P]),
    "rt_stream_wait": (C.c_int, [_P, _P]),
    "rt_resolve": (C.c_int, [_P, C.c_float]),
    "rt_ipc_alloc": (C.c_int, [C.c_int, C.c_size_t, _PP, _P]),
    "rt_ipc_open": (C.c_int, [C.c_int, _P, _PP]),
    "rt_ipc_close": (C.c_int, [C.c_int, _P]),
    "rt_ipc_free": (C.c_int, [C.c_int, _P]),
    "rt_trace_crop": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rt_get_counters": (C.c_int, [_P, C.POINTER(Counters)]),
    "rt_reset_counters": (C.c_int, [_P]),
    "rt_timer_begin": (C.c_int, [_P]),
    "rt_timer_end": (C.c_int, [_P, _F]),
    "rt_dispatch_frames_timed": (C.c_int, [_P, C.c_int]),
    "rt_timed_total": (C.c_int, [_P, _F, C.POINTER(C.c_int)]),
    "rt_set_launch": (C.c_int, [_P, C.c_int, C.c_int]),
    "rt_get_info": (C.c_int, [_P] + [C.POINTER(C.c_int)] * 4),
    "rt_get_sieve_mode": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "rt_debug_split_info": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "rt_debug_math": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int]),
    "rt_debug_sieve_audit": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_uint64)]),
    "rt_debug_cull_audit": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_uint64)]),
    "rt_debug_fail_next_launch": (C.c_int, [_P]),
    "rt_dispatch_static": (C.c_int, [C.c_int, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      _P, _P, _P]),
    "rt_last_error": (C.c_char_p, []),
    "rt_scene_load": (C.c_int, [C.c_char_p, _PP]),
    "rt_scene_parse": (C.c_int, [C.c_char_p, C.c_size_t, _PP]),
    "rt_scene_save": (C.c_int, [_P, C.c_char_p]),
    "rt_scene_free": (C.c_int, [_P]),
    "rt_scene_sphere_count": (C.c_int, [_P]),
    "rt_scene_spheres": (C.c_void_p, [_P]),
    "rt_scene_ubo": (C.c_void_p, [_P]),
    "rt_scene_materials": (C.c_void_p, [_P]),
    "rt_scene_set_orientation": (C.c_int, [_P, C.c_float, C.c_float]),
    "rt_scene_update": (C.c_int, [_P, C.c_float]),
    "rt_scene_add_acceleration": (C.c_int, [_P, C.c_float, C.c_float, C.c_float]),
    "rt_scene_get_motion": (C.c_int, [_P, _F, _F]),
    "rt_scene_reset_frame_count": (C.c_int, [_P]),
    "rt_scene_generate": (C.c_int, [C.c_int, C.c_int, C.c_uint32, _PP]),
}

STATIC_MAX_SPHERES = 4096  # RT_STATIC_MAX_SPHERES of include/rt_hip.h
ABI_VERSION = 3            # RT_ABI_VERSION of include/rt_hip.h

_lib = None


def lib() -> C.CDLL:
    """Load the shared library once; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback implementation"
            )
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.rt_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} speaks ABI version {handle.rt_abi_version()}, this binding {ABI_VERSION}: "
                              "rebuild the library (make -C raytracing_amd/csrc)")
        _lib = handle
    return _lib


def check(status: int) -> None:
    if status != 0:
        msg = lib().rt_last_error()
        raise RtError(status, msg.decode("utf-8", "replace") if msg else "")
