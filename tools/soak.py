#!/usr/bin/env python3
"""Soak test (development aid): hundreds of dispatches with random image sizes, kernels, launch
geometries and frame counts on one context; every configuration is rendered twice, possibly with
other work in between, and must give the same bits both times and across kernels."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
scenes = [rt.Scene.default(), rt.Scene.generate(rt.KIND_RTIOW, 0, 7), rt.Scene.generate(rt.KIND_CUBE, 700, 3),
          rt.Scene.generate(rt.KIND_CUBE, 1500, 4), rt.Scene.generate(rt.KIND_CUBE, 33, 5),
          rt.Scene.generate(rt.KIND_CUBE, 4500, 6)]
sizes = [(64, 40), (256, 144), (640, 360), (1280, 720), (1000, 24), (72, 1096)]
seen = {}
t0 = time.time()
with rt.Renderer(64, 64) as r:
    r.set_max_depth(10)
    cur_scene = None
    for it in range(iters):
        if it % 50 == 0:
            print(f"iteration {it}, {time.time() - t0:.0f} s", flush=True)
        si = int(rng.integers(len(scenes)))
        w, h = sizes[int(rng.integers(len(sizes)))]
        frames = int(rng.choice([1, 2, 3, 33, 64]))  # 16 frames and more: the frames of heavy tiles shared among waves on small scenes
        kernel = int(rng.choice([0, 2, 3, 4, 1, 5, 6, 7, 8]))
        launch = [(0, 0), (1, 64), (37, 128), (512, 256), (2048, 64), (256, 1024)][int(rng.integers(6))]
        if (r.width, r.height) != (w, h):
            r.resize(w, h)
        if cur_scene != si:
            r.set_scene(scenes[si], frame_count=1)
            cur_scene = si
        r.set_kernel(kernel)
        r.set_launch(*launch)
        r.dispatch(frames)
        if rng.random() < 0.5:
            r.dispatch(frames)  # back-to-back launches on the stream (frame 1 clears again)
        acc = r.read_accum()
        key = (si, w, h, frames)
        if key in seen:
            assert np.array_equal(acc.view(np.uint32), seen[key].view(np.uint32)), (it, key, kernel, launch)
        else:
            seen[key] = acc
print(f"soak ok: {iters} iterations, {len(seen)} configurations, {time.time() - t0:.1f} s")
