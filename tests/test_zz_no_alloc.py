"""GPU test of rt_dispatch_static, the entry point that allocates nothing: called through the
C ABI with caller-owned device buffers, its images and ray count must equal, bit for bit, those of
the context path (rt_create + rt_upload_spheres + rt_dispatch_frames) on the same inputs -- on two
calls in a row, the second with a smaller scene and a smaller image, so that whatever the first
call left in the library's static device array would show."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Hip:
    """The HIP runtime the library itself is linked against, through ctypes: the test owns its
    device buffers without bringing a second runtime (PyTorch ships its own) into the process."""

    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so.7")  # already loaded by libraytracing_amd.so: same instance
        for name in ("hipMalloc", "hipFree", "hipMemcpy", "hipDeviceSynchronize", "hipMemGetInfo", "hipMemset"):
            getattr(self.rt, name).restype = C.c_int

    def ok(self, status):
        assert status == 0, f"HIP error {status}"

    def upload(self, host: np.ndarray) -> C.c_void_p:
        host = np.ascontiguousarray(host)
        ptr = C.c_void_p()
        self.ok(self.rt.hipMalloc(C.byref(ptr), C.c_size_t(max(host.nbytes, 16))))
        self.ok(self.rt.hipMemcpy(ptr, host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes), 1))
        return ptr

    def download(self, ptr, like: np.ndarray) -> np.ndarray:
        out = np.empty_like(like)
        self.ok(self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), ptr, C.c_size_t(out.nbytes), 2))
        return out

    def free(self, *ptrs):
        for p in ptrs:
            self.ok(self.rt.hipFree(p))

    def sync(self):
        self.ok(self.rt.hipDeviceSynchronize())

    def free_bytes(self) -> int:
        free, total = C.c_size_t(), C.c_size_t()
        self.ok(self.rt.hipMemGetInfo(C.byref(free), C.byref(total)))
        return free.value


def _static_render(rt, hip, scene, ubo, w, h, n_frames, depth, profile, check_free):
    lib = rt._native.lib()
    acc0 = np.full((h, w, 4), 0.5, np.float32)  # a known previous accumulation (frame 1 clears it)
    col0 = np.full((h, w, 4), 0x55, np.uint8)
    spheres = hip.upload(np.ascontiguousarray(scene.spheres).view(np.uint8))
    accum, color = hip.upload(acc0), hip.upload(col0)
    rays = hip.upload(np.full(1, -1, np.int64))
    hip.sync()
    free_before = hip.free_bytes()
    rc = lib.rt_dispatch_static(0, None, spheres, len(scene.spheres), ubo.tobytes(), w, h, n_frames, depth, profile,
                                accum, color, rays)
    rt._native.check(rc)
    hip.sync()
    # (not on a process's first call: the runtime loads the library's code object on first use)
    if check_free:
        assert hip.free_bytes() == free_before, "the call changed the amount of free device memory"
    out = hip.download(accum, acc0), hip.download(color, col0), int(hip.download(rays, np.zeros(1, np.int64))[0])
    hip.free(spheres, accum, color, rays)
    return out


def _context_render(rt, scene, ubo, w, h, n_frames, depth, profile):
    with rt.Renderer(w, h) as r:
        r.set_profile(profile)
        r.set_max_depth(depth)
        r.upload_spheres(scene.spheres)
        r.set_uniforms(ubo)
        r.write_accum(np.full((h, w, 4), 0.5, np.float32))  # the same previous accumulation
        r.dispatch(n_frames)
        return r.read_accum(), r.read_color(), r.counters()["rays"]


def test_static_dispatch_equals_context_dispatch_twice(default_scene, rtiow_scene):
    import raytracing_amd as rt

    hip = _Hip()

    cases = [
        # 484 spheres, ragged last chunk, image with a remainder column and row, both float profiles
        (rtiow_scene, 333, 190, 3, 50, rt.PROFILE_STRICT, 1),
        # then smaller: 40 spheres, smaller image, accumulating onto a previous frame (frame 2: no clear)
        (default_scene, 136, 72, 2, 8, rt.PROFILE_FUSED, 2),
        (default_scene, 64, 40, 1, 50, rt.PROFILE_STRICT, 1),
    ]
    for k, (scene, w, h, n_frames, depth, profile, frame0) in enumerate(cases):
        ubo = scene.camera.copy()
        ubo["frameCount"][0] = frame0
        acc_s, col_s, rays_s = _static_render(rt, hip, scene, ubo, w, h, n_frames, depth, profile, k > 0)
        acc_c, col_c, rays_c = _context_render(rt, scene, ubo, w, h, n_frames, depth, profile)
        hh, ww = h // 8 * 8, w // 8 * 8
        assert np.array_equal(acc_s[:hh, :ww].view(np.uint32), acc_c[:hh, :ww].view(np.uint32))
        assert np.array_equal(col_s[:hh, :ww], col_c[:hh, :ww])
        assert rays_s == rays_c and rays_s >= hh * ww * n_frames
        # texels outside floor(W/8) x floor(H/8) tiles stay untouched (compute_pipeline.cpp:429-432)
        assert (acc_s[hh:] == 0.5).all() and (acc_s[:, ww:] == 0.5).all()
        assert (col_s[hh:] == 0x55).all() and (col_s[:, ww:] == 0x55).all()


def test_static_dispatch_rejects_what_it_cannot_hold(default_scene):
    import raytracing_amd as rt

    lib = rt._native.lib()
    hip = _Hip()
    ubo = default_scene.camera.copy().tobytes()
    nmax = rt._native.STATIC_MAX_SPHERES
    buf = hip.upload(np.zeros(64 * 64 * 4, np.float32))
    big = hip.upload(np.zeros((nmax + 1) * 8, np.float32))
    try:
        assert lib.rt_dispatch_static(0, None, big, nmax + 1, ubo, 64, 64, 1, 50, 0, buf, buf, None) == -1
        assert b"at most" in lib.rt_last_error()
        # sphereCount (40) beyond the spheres passed; a null image; zero frames; an unknown device
        assert lib.rt_dispatch_static(0, None, big, 8, ubo, 64, 64, 1, 50, 0, buf, buf, None) == -1
        assert b"sphereCount" in lib.rt_last_error()
        assert lib.rt_dispatch_static(0, None, big, 40, ubo, 64, 64, 1, 50, 0, None, buf, None) == -1
        assert lib.rt_dispatch_static(0, None, big, 40, ubo, 64, 64, 0, 50, 0, buf, buf, None) == -1
        assert lib.rt_dispatch_static(99, None, big, 40, ubo, 64, 64, 1, 50, 0, buf, buf, None) == -1
        # and the largest scene it does hold goes through
        ubo_max = default_scene.camera.copy()
        ubo_max["sphereCount"][0] = nmax
        assert lib.rt_dispatch_static(0, None, big, nmax, ubo_max.tobytes(), 64, 64, 1, 2, 0, buf, buf, None) == 0
        hip.sync()
    finally:
        hip.free(buf, big)
