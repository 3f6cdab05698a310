#!/usr/bin/env python3
"""A few timed dispatches of one workload (development aid for counter collection):
   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -- python3 tools/one_dispatch.py [scene W H frames depth kernel reps sphereCount frame0]
scene: rtiow | default | cubeN.  The first three dispatches are a warm-up; the counters printed are those
of the `reps` timed dispatches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt  # noqa: E402

a = sys.argv[1:]
if a and a[0].isdigit():  # old form: one_dispatch.py KERNEL
    a = ["rtiow", "3840", "2160", "8", "50", a[0]]
name = a[0] if len(a) > 0 else "rtiow"
w = int(a[1]) if len(a) > 1 else 3840
h = int(a[2]) if len(a) > 2 else 2160
frames = int(a[3]) if len(a) > 3 else 8
depth = int(a[4]) if len(a) > 4 else 50
kernel = int(a[5]) if len(a) > 5 else 0
reps = int(a[6]) if len(a) > 6 else 1
count = int(a[7]) if len(a) > 7 else -1   # override ubo.sphereCount
frame0 = int(a[8]) if len(a) > 8 else 1   # ubo.frameCount of the first frame (1 clears: no accumulation read)
if name == "rtiow":
    scene = rt.Scene.generate(rt.KIND_RTIOW, 0, 7)
elif name == "default":
    scene = rt.Scene.default()
else:
    scene = rt.Scene.generate(rt.KIND_CUBE, int(name.replace("cube", "")), 3)
with rt.Renderer(w, h) as r:
    r.set_kernel(kernel)
    r.set_max_depth(depth)
    r.set_scene(scene, frame_count=frame0)
    if count >= 0:
        ubo = scene.camera.copy()
        ubo["sphereCount"][0] = count
        ubo["frameCount"][0] = frame0
        r.set_uniforms(ubo)
    for _ in range(3):  # the first leaves tile costs, the second sorts by them and allocates what a split needs
        r.dispatch(frames)
    r.sync()
    r.reset_counters()
    r.timer_begin()
    for _ in range(reps):
        r.dispatch(frames)
    ms = r.timer_end() / reps
    c = r.counters()
    print(f"{name} {w}x{h} x{frames} depth {depth} kernel {kernel}: {ms * 1e3:.1f} us/dispatch, "
          f"{c['rays'] / reps / ms / 1e6:.3f} Grays/s, rays/dispatch {c['rays'] // reps}, launch {r.info()}")
