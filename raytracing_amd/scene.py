"""Scene: Python view of the host scene model (rt_scene in include/rt_hip.h).

Mirrors the reference's Scene class (includes/scene.hpp:100-129): ``spheres`` and
``camera`` are live views of m_spheres / m_camera in their GPU layouts, ``update``
is the camera integrator, ``save`` writes the reference's YAML layout.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native as N

# 32-byte std430 sphere (includes/scene.hpp:31-36)
SPHERE_DTYPE = np.dtype(
    [("center", "<f4", 3), ("radius", "<f4"), ("color", "<f4", 3), ("padding", "<f4")]
)
# 80-byte std140 uniform block (includes/scene.hpp:22-29)
UBO_DTYPE = np.dtype(
    {
        "names": ["camera_forward", "camera_right", "camera_up", "camera_position", "sphereCount", "frameCount"],
        "formats": [("<f4", 3), ("<f4", 3), ("<f4", 3), ("<f4", 3), "<i4", "<u4"],
        "offsets": [0, 16, 32, 48, 60, 64],
        "itemsize": 80,
    }
)
assert SPHERE_DTYPE.itemsize == 32

DEFAULT_SCENE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes", "default40.yaml")

KIND_RTIOW = 0
KIND_CUBE = 1
KIND_RTIOW_MATERIALS = 2

# material extension: kind 0 = the reference's surface response
MATERIAL_REFERENCE, MATERIAL_LAMBERTIAN, MATERIAL_METAL, MATERIAL_DIELECTRIC = 0, 1, 2, 3


class _SceneView(np.ndarray):
    """A view of memory owned by a Scene: carries a reference to it (and so do its slices), so the
    native scene is not freed while some array still looks at its spheres or camera."""

    _owner = None

    def __array_finalize__(self, obj):
        if obj is not None:
            self._owner = getattr(obj, "_owner", None)


def _owned(arr: np.ndarray, owner) -> np.ndarray:
    view = arr.view(_SceneView)
    view._owner = owner
    return view


class Scene:
    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)

    # -- construction ---------------------------------------------------
    @classmethod
    def load(cls, path: str = DEFAULT_SCENE) -> "Scene":
        out = C.c_void_p()
        N.check(N.lib().rt_scene_load(os.fsencode(path), C.byref(out)))
        return cls(out.value)

    @classmethod
    def parse(cls, text: str | bytes) -> "Scene":
        data = text.encode() if isinstance(text, str) else text
        out = C.c_void_p()
        N.check(N.lib().rt_scene_parse(data, len(data), C.byref(out)))
        return cls(out.value)

    @classmethod
    def generate(cls, kind: int, count: int = 0, seed: int = 1) -> "Scene":
        out = C.c_void_p()
        N.check(N.lib().rt_scene_generate(kind, count, seed, C.byref(out)))
        return cls(out.value)

    @classmethod
    def default(cls) -> "Scene":
        """The default scene as the application shows it at launch: the first 40 spheres of
        the reference's scene file, its camera position, and the yaw=90/pitch=0 basis that
        Application::Application installs (src/application.cpp:10-23)."""
        s = cls.load(DEFAULT_SCENE)
        s.set_orientation(90.0, 0.0)
        return s

    def close(self) -> None:
        if self._h:
            N.lib().rt_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- views ----------------------------------------------------------
    @property
    def spheres(self) -> np.ndarray:
        n = N.lib().rt_scene_sphere_count(self._h)
        if n <= 0:
            return np.zeros(0, SPHERE_DTYPE)
        ptr = N.lib().rt_scene_spheres(self._h)
        buf = (C.c_char * (n * 32)).from_address(ptr)
        return _owned(np.frombuffer(buf, dtype=SPHERE_DTYPE, count=n), self)

    @property
    def camera(self) -> np.ndarray:
        ptr = N.lib().rt_scene_ubo(self._h)
        buf = (C.c_char * 80).from_address(ptr)
        return _owned(np.frombuffer(buf, dtype=UBO_DTYPE, count=1), self)

    @property
    def materials(self) -> np.ndarray:
        """Material extension: (n, 2) float32 view of {kind, param} per sphere (all zeros = reference)."""
        n = N.lib().rt_scene_sphere_count(self._h)
        ptr = N.lib().rt_scene_materials(self._h)
        if n <= 0 or not ptr:
            return np.zeros((0, 2), np.float32)
        buf = (C.c_float * (n * 2)).from_address(ptr)
        return _owned(np.frombuffer(buf, dtype=np.float32, count=n * 2).reshape(n, 2), self)

    @property
    def sphere_count(self) -> int:
        return int(self.camera["sphereCount"][0])

    @property
    def frame_count(self) -> int:
        return int(self.camera["frameCount"][0])

    @frame_count.setter
    def frame_count(self, v: int) -> None:
        self.camera["frameCount"][0] = v

    # -- behaviour ------------------------------------------------------
    def set_orientation(self, yaw_deg: float, pitch_deg: float) -> None:
        N.check(N.lib().rt_scene_set_orientation(self._h, yaw_deg, pitch_deg))

    def update(self, dt: float) -> None:
        N.check(N.lib().rt_scene_update(self._h, dt))

    def add_acceleration(self, ax: float, ay: float, az: float) -> None:
        N.check(N.lib().rt_scene_add_acceleration(self._h, ax, ay, az))

    def motion(self):
        a = (C.c_float * 3)()
        v = (C.c_float * 3)()
        N.check(N.lib().rt_scene_get_motion(self._h, a, v))
        return np.array(a, np.float32), np.array(v, np.float32)

    def reset_frame_count(self) -> None:
        N.check(N.lib().rt_scene_reset_frame_count(self._h))

    def save(self, path: str) -> None:
        N.check(N.lib().rt_scene_save(self._h, os.fsencode(path)))
