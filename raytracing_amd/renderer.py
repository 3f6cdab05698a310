"""Renderer: Python view of the HIP backend (rt_context in include/rt_hip.h).

Stands where the reference's ComputePipeline stands
(src/engine/src/compute_pipeline.cpp): it owns the accumulation and colour
images, takes the scene's sphere buffer and uniform block, and dispatches the
kernel.  Everything runs in libraytracing_amd.so on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .scene import Scene

PROFILE_STRICT = 0
PROFILE_FUSED = 1
KERNEL_STREAM = 0
KERNEL_TILE = 1
KERNEL_STREAM_SGPR = 2
KERNEL_STREAM_LDS = 3
KERNEL_STREAM_CULLED = 4
KERNEL_STREAM_MFMA = 5
KERNEL_LITE = 6
KERNEL_BVH = 8
CLEAR_REFERENCE = 0
CLEAR_FIRST = 1
CLEAR_NEVER = 2
OPTION_CAMERA_SHORTCUTS = 0

RECORD_DTYPE = np.dtype([("hit", "<i4"), ("hash", "<u4", 3)])


class Renderer:
    def __init__(self, width: int, height: int, device: int = 0):
        out = C.c_void_p()
        N.check(N.lib().rt_create(device, width, height, C.byref(out)))
        self._h = out
        self.width, self.height = width, height
        self.max_depth = 50

    def close(self) -> None:
        if self._h:
            N.lib().rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- scene upload ---------------------------------------------------
    def upload_spheres(self, spheres: np.ndarray) -> None:
        arr = np.ascontiguousarray(spheres)
        if arr.dtype.itemsize != 32 and arr.size:
            raise ValueError("spheres must be 32-byte records")
        ptr = arr.ctypes.data_as(C.c_void_p) if arr.size else None
        N.check(N.lib().rt_upload_spheres(self._h, ptr, int(arr.shape[0]) if arr.size else 0))

    def upload_materials(self, materials: np.ndarray | None) -> None:
        """Material extension: (n, 2) float32 {kind, param} per uploaded sphere; None switches it off."""
        if materials is None or len(materials) == 0:
            N.check(N.lib().rt_upload_materials(self._h, None, 0))
            return
        arr = np.ascontiguousarray(materials, np.float32).reshape(-1, 2)
        N.check(N.lib().rt_upload_materials(self._h, arr.ctypes.data_as(C.c_void_p), arr.shape[0]))

    def set_uniforms(self, ubo: np.ndarray | bytes) -> None:
        raw = ubo.tobytes() if isinstance(ubo, np.ndarray) else bytes(ubo)
        if len(raw) != 80:
            raise ValueError("uniform block must be 80 bytes")
        N.check(N.lib().rt_set_uniforms(self._h, raw))

    def set_scene(self, scene: Scene, frame_count: int | None = None) -> None:
        """updateScene (compute_pipeline.cpp:491-496): spheres + uniform block (+ materials, if any)."""
        self.upload_spheres(scene.spheres)
        mats = scene.materials
        self.upload_materials(mats if len(mats) and mats[:, 0].any() else None)
        cam = scene.camera.copy()
        if frame_count is not None:
            cam["frameCount"][0] = frame_count
        self.set_uniforms(cam)

    # -- parameters -----------------------------------------------------
    def set_max_depth(self, depth: int) -> None:
        N.check(N.lib().rt_set_max_depth(self._h, depth))
        self.max_depth = depth

    def set_profile(self, profile: int) -> None:
        N.check(N.lib().rt_set_profile(self._h, profile))

    def set_kernel(self, kernel: int) -> None:
        N.check(N.lib().rt_set_kernel(self._h, kernel))

    def set_option(self, option: int, value: int) -> None:
        N.check(N.lib().rt_set_option(self._h, option, value))

    def set_camera_shortcuts(self, on: bool) -> None:
        """False: every Trace() call evaluates the whole sphere list (RT_OPTION_CAMERA_SHORTCUTS)."""
        self.set_option(OPTION_CAMERA_SHORTCUTS, 1 if on else 0)

    def set_tile_rows(self, row0: int, row1: int) -> None:
        N.check(N.lib().rt_set_tile_rows(self._h, row0, row1))

    def set_frame_schedule(self, stride: int = 1, clear_mode: int = CLEAR_REFERENCE) -> None:
        N.check(N.lib().rt_set_frame_schedule(self._h, stride, clear_mode))

    def set_launch(self, blocks: int = 0, threads: int = 0) -> None:
        N.check(N.lib().rt_set_launch(self._h, blocks, threads))

    def resize(self, width: int, height: int) -> None:
        N.check(N.lib().rt_resize(self._h, width, height))
        self.width, self.height = width, height

    # -- execution ------------------------------------------------------
    def dispatch(self, n_frames: int = 1) -> None:
        N.check(N.lib().rt_dispatch_frames(self._h, n_frames))

    def dispatch_timed(self, n_frames: int = 1) -> None:
        """dispatch() bracketed by an event pair of its own; timed_total() sums them."""
        N.check(N.lib().rt_dispatch_frames_timed(self._h, n_frames))

    def timed_total(self):
        """(device milliseconds, dispatches) of the dispatch_timed calls since the last call; synchronises."""
        ms, n = C.c_float(), C.c_int()
        N.check(N.lib().rt_timed_total(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def sync(self) -> None:
        N.check(N.lib().rt_sync(self._h))

    def resolve(self, divisor: float) -> None:
        N.check(N.lib().rt_resolve(self._h, divisor))

    def timer_begin(self) -> None:
        N.check(N.lib().rt_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_float()
        N.check(N.lib().rt_timer_end(self._h, C.byref(ms)))
        return float(ms.value)

    # -- results --------------------------------------------------------
    def read_color(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.uint8)
        N.check(N.lib().rt_read_color(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def read_accum(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.float32)
        N.check(N.lib().rt_read_accum(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def write_accum(self, accum: np.ndarray) -> None:
        arr = np.ascontiguousarray(accum, np.float32)
        if arr.shape != (self.height, self.width, 4):
            raise ValueError("accum must be H x W x 4 float32")
        N.check(N.lib().rt_write_accum(self._h, arr.ctypes.data_as(C.c_void_p)))

    def device_buffers(self):
        a, c = C.c_void_p(), C.c_void_p()
        N.check(N.lib().rt_device_buffers(self._h, C.byref(a), C.byref(c)))
        return a.value, c.value

    def bind_buffers(self, accum_ptr: int | None, color_ptr: int | None) -> None:
        N.check(N.lib().rt_bind_buffers(self._h, C.c_void_p(accum_ptr), C.c_void_p(color_ptr)))

    def bind_buffers_next(self, accum_ptr: int = 0, color_ptr: int = 0) -> None:
        """As bind_buffers without waiting: the images take effect with the next dispatch."""
        N.check(N.lib().rt_bind_buffers_next(self._h, C.c_void_p(accum_ptr), C.c_void_p(color_ptr)))

    def sync_dispatch(self, back: int = 0) -> None:
        """Wait for the dispatch issued `back` dispatches ago (0: the last one, 1: the one before)."""
        N.check(N.lib().rt_sync_dispatch(self._h, back))

    def wait_stream(self, hip_stream: int | None) -> None:
        """Dispatches issued after this call start only when the work queued on `hip_stream` (a raw
        hipStream_t, e.g. torch.cuda.current_stream().cuda_stream) so far has finished."""
        N.check(N.lib().rt_wait_stream(self._h, C.c_void_p(hip_stream)))

    def stream_wait(self, hip_stream: int | None) -> None:
        """Work queued on `hip_stream` after this call starts only when the dispatches issued so far have finished."""
        N.check(N.lib().rt_stream_wait(self._h, C.c_void_p(hip_stream)))

    def trace_crop(self, n_frames: int, x0: int, y0: int, w: int, h: int, want_light: bool = True):
        """Records (n_frames, h, w, max_depth) and per-frame light (n_frames, h, w, 3)."""
        rec = np.empty((n_frames, h, w, self.max_depth), RECORD_DTYPE)
        light = np.empty((n_frames, h, w, 3), np.float32) if want_light else None
        N.check(
            N.lib().rt_trace_crop(
                self._h, n_frames, x0, y0, w, h,
                rec.ctypes.data_as(C.c_void_p),
                light.ctypes.data_as(C.c_void_p) if want_light else None,
            )
        )
        return rec, light

    def counters(self) -> dict:
        c = N.Counters()
        N.check(N.lib().rt_get_counters(self._h, C.byref(c)))
        return {"rays": int(c.rays), "sphere_tests": int(c.sphere_tests), "paths": int(c.paths), "list_rays": int(c.list_rays)}

    def reset_counters(self) -> None:
        N.check(N.lib().rt_reset_counters(self._h))

    def info(self) -> dict:
        v = [C.c_int() for _ in range(4)]
        N.check(N.lib().rt_get_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("device", "cu_count", "blocks", "threads"), (int(x.value) for x in v)))

    def sieve_mode(self) -> int:
        """Sieve of the last stream-kernel dispatch: 0/1 vector ALU, 2 culled, 3 matrix cores, -1 none."""
        m = C.c_int()
        N.check(N.lib().rt_get_sieve_mode(self._h, C.byref(m)))
        return int(m.value)

    def split_info(self) -> dict:
        """How the last stream-kernel dispatch shared the frames of heavy tiles (rt_debug_split_info)."""
        out = (C.c_int * 4)()
        N.check(N.lib().rt_debug_split_info(self._h, out))
        return dict(zip(("ways", "frames", "most_tiles", "tiles"), (int(x) for x in out)))

    def sieve_audit(self, rays: np.ndarray) -> dict:
        """rays: (n, 6) float32 origin+direction; see rt_debug_sieve_audit in include/rt_hip.h."""
        arr = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = (C.c_uint64 * 4)()
        N.check(N.lib().rt_debug_sieve_audit(self._h, arr.ctypes.data_as(C.c_void_p), arr.shape[0], out))
        return {"missed_strict": int(out[0]), "missed_fused": int(out[1]), "exact": int(out[2]), "passed": int(out[3])}

    def cull_audit(self, rays: np.ndarray) -> dict:
        """rays: (n, 6) float32 origin+direction; see rt_debug_cull_audit in include/rt_hip.h."""
        arr = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = (C.c_uint64 * 7)()
        N.check(N.lib().rt_debug_cull_audit(self._h, arr.ctypes.data_as(C.c_void_p), arr.shape[0], out))
        return {"missed": int(out[0]), "culled": int(out[1]), "tested": int(out[2]),
                "boxes_culled": int(out[3]), "boxes_tested": int(out[4]),
                "hits_pruned": int(out[5]), "hits": int(out[6])}

    def debug_math(self, op: int, a, b=None, c=None) -> np.ndarray:
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float32)
        c = np.ascontiguousarray(c if c is not None else np.zeros_like(a), np.float32)
        out = np.empty_like(a)
        N.check(
            N.lib().rt_debug_math(
                self._h, op, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                c.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), a.size,
            )
        )
        return out
