#!/usr/bin/env python3
"""Ad-hoc timing of the kernels on one GPU (development aid; bench.py is the contract).

usage: python tools/quick_bench.py [scene=rtiow|rtiowmat|default|cube4096] [W H] [frames] [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt  # noqa: E402


def main():
    a = sys.argv[1:]
    scene_name = a[0] if len(a) > 0 else "rtiow"
    w = int(a[1]) if len(a) > 1 else 3840
    h = int(a[2]) if len(a) > 2 else 2160
    frames = int(a[3]) if len(a) > 3 else 4
    reps = int(a[4]) if len(a) > 4 else 3
    if scene_name == "rtiow":
        scene = rt.Scene.generate(rt.KIND_RTIOW, 0, 7)
    elif scene_name == "rtiowmat":
        scene = rt.Scene.generate(rt.KIND_RTIOW_MATERIALS, 0, 7)
    elif scene_name == "default":
        scene = rt.Scene.default()
    else:
        scene = rt.Scene.generate(rt.KIND_CUBE, int(scene_name.replace("cube", "")), 3)
    n = scene.sphere_count
    print(f"scene {scene_name}: {n} spheres, {w}x{h}, {frames} frames/dispatch")
    ubo = scene.camera.copy()
    ubo["frameCount"][0] = 1
    with rt.Renderer(w, h) as r:
        r.set_scene(scene, frame_count=1)
        print(r.info())
        for kname, kernel in (("lds", 3), ("sgpr", 2), ("culled", 4), ("tile", 1)):
            for pname, profile in (("strict", 0), ("fused", 1)):
                r.set_kernel(kernel)
                r.set_profile(profile)
                r.dispatch(1)
                r.sync()
                best = None
                for _ in range(reps):
                    r.reset_counters()
                    r.timer_begin()
                    r.dispatch(frames)
                    ms = r.timer_end()
                    c = r.counters()
                    best = ms if best is None or ms < best else best
                rays = c["rays"]
                tests = c["sphere_tests"]
                print(f"{kname:6s} {pname:6s}: {best:9.3f} ms  {rays / best / 1e6:8.3f} Grays/s  "
                      f"{tests / best / 1e9:7.3f} Ttests/s  {tests * 21 / best / 1e9:7.2f} TFLOP/s(alg)  "
                      f"rays/path {rays / c['paths']:.2f}")


if __name__ == "__main__":
    main()
