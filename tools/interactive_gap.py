#!/usr/bin/env python3
"""Frame rate of one-frame 1080p dispatches of the default scene under different host loops
(development aid): host sync every frame; two frames in flight; the same with per-dispatch timing
events; no synchronisation at all (the GPU's own back-to-back rate)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt  # noqa: E402

scene = rt.Scene.default()
n = 2000
with rt.Renderer(1920, 1080) as r:
    r.set_scene(scene, frame_count=1)
    ubo = scene.camera.copy()

    def run(name, body):
        for _ in range(50):
            body(0)
        r.sync()
        t0 = time.perf_counter()
        for f in range(n):
            body(f)
        r.sync()
        dt = time.perf_counter() - t0
        print(f"{name:48s} {dt / n * 1e6:7.1f} us/frame")

    def uniforms(f):
        ubo["frameCount"][0] = 2 + f % 30
        r.set_uniforms(ubo)

    def a(f):
        uniforms(f); r.dispatch(1); r.sync()

    def b(f):
        uniforms(f); r.dispatch(1); r.sync_dispatch(1)

    def c(f):
        uniforms(f); r.dispatch_timed(1); r.sync_dispatch(1)

    def d(f):
        uniforms(f); r.dispatch(1)

    run("host sync every frame", a)
    run("two frames in flight", b)
    run("two frames in flight, timed dispatches", c)
    r.timed_total()
    run("no synchronisation", d)
