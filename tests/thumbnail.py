"""Shared helpers of the screenshot comparison: the reference's thumbnail.png reduced 4x
(tests/golden/thumbnail_x4.npz) and the camera pose recovered for it (thumbnail_pose.json)."""
import json
import os

import numpy as np

import oracle as O
from conftest import GOLDEN

WIDTH, HEIGHT = 1906, 1013      # the screenshot's size; an 8x8-tiled dispatch covers 1904 x 1008 of it


def posed_ubo(scene):
    """The default scene's uniform block at the recovered pose, frameCount = 1."""
    pose = json.load(open(os.path.join(GOLDEN, "thumbnail_pose.json")))
    ubo = scene.camera.copy()
    f, r, u = O.camera_basis(pose["yaw_degrees"], pose["pitch_degrees"])
    ubo["camera_forward"][0], ubo["camera_right"][0], ubo["camera_up"][0] = f, r, u
    ubo["camera_position"][0] = pose["position"]
    ubo["frameCount"][0] = 1
    return ubo


def reduced_screenshot():
    return np.load(os.path.join(GOLDEN, "thumbnail_x4.npz"))["rgb"].astype(np.float64)


def box(a, k):
    h, w = a.shape[0] // k * k, a.shape[1] // k * k
    return a[:h, :w].reshape(h // k, k, w // k, k, -1).mean((1, 3))


def psnr(a, b):
    return 10.0 * np.log10(255.0 ** 2 / np.mean((a - b) ** 2))


def compare(accum, n_frames):
    """accum: H x W x 4 float accumulation after n_frames -> (PSNR at 4x, PSNR at 8x, sky agreement)."""
    img = np.clip(accum[:1008, :1904, :3] / np.float32(n_frames), 0.0, 1.0).astype(np.float64) * 255.0
    ref = reduced_screenshot()
    mine = box(img, 4)
    sky = np.array([23.0, 27.0, 34.0])
    ref_sky = np.all(ref == sky, axis=-1)
    my_sky = np.all(np.abs(mine - sky) < 0.6, axis=-1)
    return psnr(mine, ref), psnr(box(mine, 2), box(ref, 2)), float(np.mean(ref_sky == my_sky))
