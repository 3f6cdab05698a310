"""ctypes binding of the CPU oracle (oracle/librt_oracle.so) for the tests.

The oracle is test infrastructure: it restates the reference shader on the CPU
and is only ever used as the checker (see oracle/rt_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "librt_oracle.so")

STRICT, FUSED = 0, 1
CLEAR_REFERENCE, CLEAR_FIRST, CLEAR_NEVER = 0, 1, 2

RECORD_DTYPE = np.dtype([("hit", "<i4"), ("hash", "<u4", 3)])


class Job(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("tile_row0", C.c_int), ("tile_row1", C.c_int),
        ("max_depth", C.c_int), ("profile", C.c_int),
        ("n_frames", C.c_int), ("frame_stride", C.c_int),
        ("clear_mode", C.c_int), ("n_threads", C.c_int),
        ("materials", C.c_void_p),
        ("const_roughness", C.c_float),
    ]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "librt_oracle.so"])
        L = C.CDLL(LIB_PATH)
        L.rto_wang_hash.restype = C.c_uint32
        L.rto_wang_hash.argtypes = [C.c_uint32]
        L.rto_seed.restype = C.c_uint32
        L.rto_seed.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32]
        L.rto_hash_to_float.restype = C.c_float
        L.rto_hash_to_float.argtypes = [C.c_uint32]
        L.rto_unorm8.restype = C.c_uint8
        L.rto_unorm8.argtypes = [C.c_float]
        L.rto_render.restype = C.c_int
        L.rto_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Job), C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_uint64)]
        L.rto_trace_crop.restype = C.c_int
        L.rto_trace_crop.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Job), C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rto_sincos_2pi.restype = None
        L.rto_sincos_2pi.argtypes = [C.c_float, C.POINTER(C.c_float)]
        L.rto_camera_basis.restype = None
        L.rto_camera_basis.argtypes = [C.c_float, C.c_float] + [C.POINTER(C.c_float)] * 3
        L.rto_scene_update.restype = None
        L.rto_scene_update.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float]
        _lib = L
    return _lib


def wang_hash(x: int) -> int:
    return int(lib().rto_wang_hash(x & 0xFFFFFFFF))


def seed(px: int, py: int, frame: int, bounce: int) -> int:
    return int(lib().rto_seed(px, py, frame, bounce))


def _job(width, height, max_depth, profile, n_frames, stride, clear, rows, threads, materials=None,
         const_roughness=0.0) -> Job:
    r0, r1 = rows if rows else (0, 0)
    ptr = materials.ctypes.data if materials is not None else None
    return Job(width, height, r0, r1, max_depth, profile, n_frames, stride, clear, threads, ptr, const_roughness)


def _materials(materials, n):
    """(n, 2) float32 {kind, param} per sphere, or None for the reference's surface response."""
    if materials is None:
        return None
    m = np.ascontiguousarray(materials, np.float32).reshape(-1, 2)
    assert m.shape[0] == n
    return m


def render(ubo: np.ndarray, spheres: np.ndarray, width: int, height: int, *, max_depth=50, profile=STRICT,
           n_frames=1, frame_stride=1, clear_mode=CLEAR_REFERENCE, rows=None, threads=1, accum=None,
           materials=None, const_roughness=0.0):
    """Returns (accum HxWx4 f32, color HxWx4 u8, rays).  ubo: 80-byte record array (frameCount = first frame)."""
    ubo_b = np.ascontiguousarray(ubo).tobytes()
    assert len(ubo_b) == 80
    sph = np.ascontiguousarray(spheres)
    acc = np.zeros((height, width, 4), np.float32) if accum is None else np.ascontiguousarray(accum, np.float32).copy()
    col = np.zeros((height, width, 4), np.uint8)
    rays = C.c_uint64()
    mats = _materials(materials, len(sph))
    job = _job(width, height, max_depth, profile, n_frames, frame_stride, clear_mode, rows, threads, mats,
               const_roughness)
    rc = lib().rto_render(ubo_b, sph.ctypes.data_as(C.c_void_p), len(sph), C.byref(job),
                          acc.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), C.byref(rays))
    if rc:
        raise RuntimeError(f"rto_render failed: {rc}")
    return acc, col, int(rays.value)


def trace_crop(ubo: np.ndarray, spheres: np.ndarray, width: int, height: int, x0, y0, w, h, *, max_depth=50,
               profile=STRICT, n_frames=1, frame_stride=1, materials=None):
    """Returns (records (n_frames,h,w,max_depth), light (n_frames,h,w,3))."""
    ubo_b = np.ascontiguousarray(ubo).tobytes()
    sph = np.ascontiguousarray(spheres)
    rec = np.empty((n_frames, h, w, max_depth), RECORD_DTYPE)
    light = np.empty((n_frames, h, w, 3), np.float32)
    mats = _materials(materials, len(sph))
    job = _job(width, height, max_depth, profile, n_frames, frame_stride, CLEAR_REFERENCE, None, 1, mats)
    rc = lib().rto_trace_crop(ubo_b, sph.ctypes.data_as(C.c_void_p), len(sph), C.byref(job), x0, y0, w, h,
                              rec.ctypes.data_as(C.c_void_p), light.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError(f"rto_trace_crop failed: {rc}")
    return rec, light


def sincos_2pi(u: float):
    out = (C.c_float * 2)()
    lib().rto_sincos_2pi(u, out)
    return float(out[0]), float(out[1])


def camera_basis(yaw: float, pitch: float):
    f, r, u = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    lib().rto_camera_basis(yaw, pitch, f, r, u)
    return np.array(f, np.float32), np.array(r, np.float32), np.array(u, np.float32)


def scene_update(ubo: np.ndarray, accel: np.ndarray, vel: np.ndarray, dt: float):
    """In-place Scene::update on an 80-byte ubo record array and float32[3] accel/vel."""
    assert ubo.dtype.itemsize == 80 and accel.dtype == np.float32 and vel.dtype == np.float32
    lib().rto_scene_update(ubo.ctypes.data_as(C.c_void_p), accel.ctypes.data_as(C.POINTER(C.c_float)),
                           vel.ctypes.data_as(C.POINTER(C.c_float)), dt)
