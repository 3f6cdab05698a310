#!/usr/bin/env python3
"""Recover the camera pose of the reference's thumbnail.png (development tool, minutes of CPU).

thumbnail.png is the one rendered output the reference ships: a 1906x1013 screenshot of its default
scene at an unrecorded camera position and orientation.  This script finds the pose (position,
yaw, pitch: the five numbers the application's controls change, src/application.cpp:38-118) so
that the image becomes a golden output for the whole path:

  1. coarse: random camera poses around sphere 11 (the close-up sphere, identified by its
     one-bounce colour, see tools/make_thumbnail_fixture.py); score = disagreement of the sky
     mask (which pixels show the sky colour) on a 120x64 grid, first hits only, numpy;
  2. refine: Nelder-Mead on the same score at 476x253;
  3. refine: Nelder-Mead on the mean absolute colour difference between the 6x box-filtered
     screenshot and the CPU oracle's render (half resolution, 4 frames);
  4. refine: the same on full-size renders (3 frames), 3x box-filtered, differences capped at 96.

Result -> tests/golden/thumbnail_pose.json.  Needs /root/reference/thumbnail.png and the oracle.
usage: python tools/recover_thumbnail_pose.py [seed=1] [coarse_samples=200000]
"""
import json
import os
import sys

import numpy as np
from PIL import Image
from scipy.optimize import minimize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
import raytracing_amd as rt  # noqa: E402

THUMB = "/root/reference/thumbnail.png"
SKY = np.array([23, 27, 34], np.float32)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    samples = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    scene = rt.Scene.load()
    sph = scene.spheres[:40]
    C = sph["center"].astype(np.float64)
    R = sph["radius"].astype(np.float64)
    img = np.array(Image.open(THUMB).convert("RGB")).astype(np.float32)
    H0, W0 = img.shape[:2]
    sky = np.all(img == SKY, axis=2).astype(np.float64)

    def grid(w, h):
        yi = np.arange(H0) * h // H0
        xi = np.arange(W0) * w // W0
        tot = np.zeros((h, w))
        cnt = np.zeros((h, w))
        np.add.at(tot, (yi[:, None], xi[None, :]), sky)
        np.add.at(cnt, (yi[:, None], xi[None, :]), 1)
        X, Y = np.meshgrid((np.arange(w) + 0.5) * W0 / w, (np.arange(h) + 0.5) * H0 / h)
        return tot / cnt, X, Y

    def sky_loss(p, target, X, Y):
        f, r, u = (v.astype(np.float64) for v in O.camera_basis(float(p[3]), float(p[4])))
        hh = (2 * X - W0) / W0           # shader.comp:75-76: both divide by the width
        vv = (2 * Y - H0) / W0
        d = f[None, None] + hh[..., None] * r + vv[..., None] * u
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        L = C - np.asarray(p[:3])
        s = np.einsum("hwk,nk->hwn", d, L)
        q = s * s - (L * L).sum(1) + R * R
        hit = (q >= 0) & (s + np.sqrt(np.maximum(q, 0)) > 0)
        return np.abs((~hit.any(-1)) - target).mean()

    # 1. coarse
    rng = np.random.default_rng(seed)
    target, X, Y = grid(120, 64)
    best = (1e9, None)
    for _ in range(samples):
        v = rng.standard_normal(3)
        p = np.concatenate([C[11] + v / np.linalg.norm(v) * rng.uniform(2.0, 4.0),
                            [rng.uniform(0, 360), rng.uniform(-89, 89)]])
        e = sky_loss(p, target, X, Y)
        if e < best[0]:
            best = (e, p)
    print("coarse", best)

    # 2. sky mask at 476x253
    target, X, Y = grid(476, 253)
    x = best[1]
    for rnd in range(4):
        simplex = x + np.vstack([np.zeros(5), np.diag([0.2, 0.2, 0.2, 1.5, 1.5]) * 0.5 ** rnd])
        x = minimize(lambda p: sky_loss(p, target, X, Y), x, method="Nelder-Mead",
                     options=dict(xatol=1e-4, fatol=1e-6, maxiter=1500, initial_simplex=simplex)).x
    print("sky mask", sky_loss(x, target, X, Y), x)

    # 3. colour, oracle at half resolution
    W, H = 952, 504
    ref = np.array(Image.open(THUMB).convert("RGB").crop((0, 0, 1904, 1008)).resize((W, H), Image.BOX)).astype(np.float64)

    def down(a, k):
        return a[: a.shape[0] // k * k, : a.shape[1] // k * k].reshape(a.shape[0] // k, k, a.shape[1] // k, k, -1).mean((1, 3))

    refd = down(ref, 6)

    def colour_loss(p):
        ubo = scene.camera.copy()
        f, r, u = O.camera_basis(float(p[3]), float(p[4]))
        ubo["camera_forward"][0], ubo["camera_right"][0], ubo["camera_up"][0] = f, r, u
        ubo["camera_position"][0] = p[:3]
        ubo["frameCount"][0] = 1
        acc, _, _ = O.render(ubo, scene.spheres, W, H, n_frames=4, threads=os.cpu_count() or 1, max_depth=12)
        return np.abs(down(np.clip(acc[..., :3] / 4, 0, 1) * 255, 6) - refd).mean()

    for rnd in range(3):
        simplex = x + np.vstack([np.zeros(5), np.diag([0.06, 0.06, 0.06, 0.4, 0.4]) * 0.5 ** rnd])
        x = minimize(colour_loss, x, method="Nelder-Mead",
                     options=dict(xatol=1e-5, fatol=1e-5, maxiter=200, initial_simplex=simplex)).x
    print("colour", colour_loss(x), x)

    # 4. colour, full size
    full = np.array(Image.open(THUMB).convert("RGB")).astype(np.float64)[:1008, :1904]
    fulld = down(full, 3)

    def full_loss(p):
        ubo = scene.camera.copy()
        f, r, u = O.camera_basis(float(p[3]), float(p[4]))
        ubo["camera_forward"][0], ubo["camera_right"][0], ubo["camera_up"][0] = f, r, u
        ubo["camera_position"][0] = p[:3]
        ubo["frameCount"][0] = 1
        acc, _, _ = O.render(ubo, scene.spheres, W0, H0, n_frames=3, threads=os.cpu_count() or 1, max_depth=16)
        img = np.clip(acc[:1008, :1904, :3] / 3, 0, 1) * 255
        return np.minimum(np.abs(down(img, 3) - fulld), 96).mean()

    for rnd in range(2):
        simplex = x + np.vstack([np.zeros(5), np.diag([0.01, 0.01, 0.01, 0.08, 0.08]) * 0.5 ** rnd])
        x = minimize(full_loss, x, method="Nelder-Mead",
                     options=dict(xatol=1e-6, fatol=1e-6, maxiter=150, initial_simplex=simplex)).x
    print("full size", full_loss(x), x)
    out = {
        "source": "camera pose of the reference's thumbnail.png recovered by tools/recover_thumbnail_pose.py",
        "image_size": [W0, H0],
        "position": [float(v) for v in x[:3]],
        "yaw_degrees": float(x[3]),
        "pitch_degrees": float(x[4]),
    }
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "thumbnail_pose.json"), "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
