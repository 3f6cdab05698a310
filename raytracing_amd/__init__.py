"""raytracing_amd: MI355X (gfx950) backend for the sphere path-tracer's compute kernel.

The work is done by libraytracing_amd.so (HIP kernels behind the C ABI of
include/rt_hip.h); this package is the thin Python host used by the tests and
the benchmark.
"""
from .scene import (  # noqa: F401
    Scene, SPHERE_DTYPE, UBO_DTYPE, DEFAULT_SCENE, KIND_RTIOW, KIND_CUBE, KIND_RTIOW_MATERIALS,
    MATERIAL_REFERENCE, MATERIAL_LAMBERTIAN, MATERIAL_METAL, MATERIAL_DIELECTRIC,
)
from .renderer import (  # noqa: F401
    Renderer, RECORD_DTYPE, PROFILE_STRICT, PROFILE_FUSED, KERNEL_STREAM, KERNEL_TILE, KERNEL_STREAM_SGPR, KERNEL_STREAM_LDS,
    KERNEL_STREAM_CULLED, KERNEL_STREAM_MFMA, KERNEL_LITE, KERNEL_BVH,
    CLEAR_REFERENCE, CLEAR_FIRST, CLEAR_NEVER, OPTION_CAMERA_SHORTCUTS,
)
from ._native import RtError  # noqa: F401
