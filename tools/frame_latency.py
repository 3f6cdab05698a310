#!/usr/bin/env python3
"""Device time of one-frame dispatches (development aid): default scene, W x H, back-to-back
dispatches of `frames` frame(s) on the stream, for a list of launch geometries.

usage: python tools/frame_latency.py [W H] [frames] [depth] [quick]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt  # noqa: E402


def main():
    a = sys.argv[1:]
    w = int(a[0]) if len(a) > 0 else 1920
    h = int(a[1]) if len(a) > 1 else 1080
    frames = int(a[2]) if len(a) > 2 else 1
    depth = int(a[3]) if len(a) > 3 else 50
    quick = len(a) > 4
    scene = rt.Scene.default()
    reps = 200
    with rt.Renderer(w, h) as r:
        r.set_max_depth(depth)
        r.set_scene(scene, frame_count=1)
        launches = [(0, 0), (512, 256), (768, 256), (1024, 256)] if quick else [(0, 0), (256, 256), (512, 256), (768, 256), (1024, 256), (2048, 256), (1024, 128), (2048, 128),
                       (4096, 64), (512, 512), (256, 1024)]
        for launch in launches:
            r.set_launch(*launch)
            for _ in range(5):
                r.dispatch(frames)
            r.sync()
            r.reset_counters()
            r.timer_begin()
            for i in range(reps):
                r.dispatch(frames)
            ms = r.timer_end()
            c = r.counters()
            print(f"launch {launch[0]:5d} x {launch[1]:4d}: {ms / reps * 1e3:8.1f} us/dispatch  "
                  f"{c['rays'] / ms / 1e6:7.2f} Grays/s  rays/path {c['rays'] / c['paths']:.2f}", flush=True)


if __name__ == "__main__":
    main()
