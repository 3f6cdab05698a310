#!/usr/bin/env python3
"""Sweep the stream kernel's launch geometry on one GPU (development aid).

usage: python tools/launch_sweep.py [scene=rtiow|default|cubeN] [W H] [frames] [kernels=2,3] [threads=64,...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt  # noqa: E402


def main():
    a = sys.argv[1:]
    name = a[0] if a else "rtiow"
    w = int(a[1]) if len(a) > 1 else 3840
    h = int(a[2]) if len(a) > 2 else 2160
    frames = int(a[3]) if len(a) > 3 else 8
    kernels = [int(k) for k in a[4].split(",")] if len(a) > 4 else [2, 3]
    thread_list = [int(k) for k in a[5].split(",")] if len(a) > 5 else [64, 128, 256, 512, 1024]
    if name == "rtiow":
        scene = rt.Scene.generate(rt.KIND_RTIOW, 0, 7)
    elif name == "default":
        scene = rt.Scene.default()
    else:
        scene = rt.Scene.generate(rt.KIND_CUBE, int(name.replace("cube", "")), 3)
    ubo = scene.camera.copy()
    ubo["frameCount"][0] = 1
    with rt.Renderer(w, h) as r:
        r.upload_spheres(scene.spheres)
        r.set_uniforms(ubo)
        cus = r.info()["cu_count"]
        for kernel in kernels:
            for threads in thread_list:
                for waves_per_cu in (4, 6, 8, 12, 16, 20, 24, 28, 32):
                    blocks = cus * waves_per_cu * 64 // threads
                    if blocks * threads // cus != waves_per_cu * 64:
                        continue
                    r.set_kernel(kernel)
                    try:
                        r.set_launch(blocks, threads)
                        r.dispatch(1)
                        r.sync()
                    except rt.RtError as e:
                        print(kernel, threads, waves_per_cu, "skip:", e.message[:60])
                        continue
                    best = 1e9
                    for _ in range(3):
                        r.reset_counters()
                        r.timer_begin()
                        r.dispatch(frames)
                        best = min(best, r.timer_end())
                    rays = r.counters()["rays"]
                    print(f"kernel {kernel} threads {threads:4d} waves/CU {waves_per_cu:2d} blocks {blocks:5d}: "
                          f"{best:8.3f} ms {rays / best / 1e6:7.3f} Grays/s", flush=True)


if __name__ == "__main__":
    main()
