#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/ from the CPU oracle.

The reference ships no golden data and cannot run here, so these vectors pin the
ORACLE (and through it the HIP kernels), not the reference: a change in either
that moves a single bit shows up as a diff of these files.

  default_crop16.npz  per-bounce hit indices + RNG hashes + light, 16x16 crop at (604,197)
                      of the 1280x720 default scene, frames 1, 2, 64, depth 50, both profiles
  default_64x36.npz   accumulation + colour images of the default scene at 64x36 after
                      frames 1 and 64 (rows 32..35 are never rendered: 36 is not a multiple of 8)
  glm_probe.txt       what the glm headers vendored in the reference compute for the host
                      camera math (only refreshed when oracle/_ref/glm_probe has been built)

usage: python tools/make_golden.py
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
import raytracing_amd as rt  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(GOLD, exist_ok=True)
    scene = rt.Scene.default()
    sph = scene.spheres.copy()

    crop = {}
    for name, prof in (("strict", O.STRICT), ("fused", O.FUSED)):
        for frame in (1, 2, 64):
            ubo = scene.camera.copy()
            ubo["frameCount"][0] = frame
            rec, light = O.trace_crop(ubo, sph, 1280, 720, 604, 197, 16, 16, profile=prof)
            crop[f"{name}_hit_f{frame}"] = rec["hit"][0].astype(np.int16)
            crop[f"{name}_hash_f{frame}"] = rec["hash"][0]
            crop[f"{name}_light_f{frame}"] = light[0]
    np.savez_compressed(os.path.join(GOLD, "default_crop16.npz"), **crop)

    img = {}
    for name, prof in (("strict", O.STRICT), ("fused", O.FUSED)):
        ubo = scene.camera.copy()
        ubo["frameCount"][0] = 1
        acc1, col1, rays1 = O.render(ubo, sph, 64, 36, profile=prof, n_frames=1)
        acc64, col64, rays64 = O.render(ubo, sph, 64, 36, profile=prof, n_frames=64)
        img[f"{name}_accum_f1"] = acc1
        img[f"{name}_color_f1"] = col1
        img[f"{name}_accum_f64"] = acc64
        img[f"{name}_color_f64"] = col64
        img[f"{name}_rays"] = np.array([rays1, rays64], np.int64)
    np.savez_compressed(os.path.join(GOLD, "default_64x36.npz"), **img)

    probe = os.path.join(ROOT, "oracle", "_ref", "glm_probe")
    if os.path.exists(probe):
        out = subprocess.check_output([probe])
        open(os.path.join(GOLD, "glm_probe.txt"), "wb").write(out)
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
