#!/usr/bin/env python3
"""Full-size comparison with the reference's thumbnail.png (how the figures quoted for the screenshot comparison and in
tests/golden/thumbnail_pose.json were measured).  Two steps, because the screenshot exists only
where the reference tree is and the GPU only on the GPU box:

  on the GPU box:   python tools/compare_thumbnail.py render [frames=4096]
                    -> gpurun_out/pose_color.npy, gpurun_out/pose_first_hit.npy
  where /root/reference is:  python tools/compare_thumbnail.py compare
                    -> PSNR per pixel and after 2x/4x/8x box filtering, share of equal pixels,
                       the same split by "close-up sphere (11) or not", gpurun_out/thumbnail_diff.png
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "gpurun_out")


def render(frames):
    import raytracing_amd as rt
    import thumbnail as T

    scene = rt.Scene.default()
    ubo = T.posed_ubo(scene)
    os.makedirs(OUT, exist_ok=True)
    with rt.Renderer(T.WIDTH, T.HEIGHT) as r:
        r.upload_spheres(scene.spheres)
        r.set_uniforms(ubo)
        r.dispatch(frames)
        np.save(os.path.join(OUT, "pose_color.npy"), r.read_color())
        r.set_max_depth(1)
        rec, _ = r.trace_crop(1, 0, 0, 1904, 1008, want_light=False)
        np.save(os.path.join(OUT, "pose_first_hit.npy"), rec["hit"][0, :, :, 0].astype(np.int8))
    print("rendered", frames, "frames")


def compare(path="/root/reference/thumbnail.png"):
    from PIL import Image

    import thumbnail as T

    ref = np.array(Image.open(path).convert("RGB")).astype(np.float64)[:1008, :1904]
    col = np.load(os.path.join(OUT, "pose_color.npy"))[:1008, :1904, :3].astype(np.float64)
    hit = np.load(os.path.join(OUT, "pose_first_hit.npy"))

    def psnr(a, b, m=None):
        d = (a - b) ** 2
        return 10 * np.log10(255.0 ** 2 / (d[m] if m is not None else d).mean())

    d = np.abs(col - ref).max(-1)
    for name, m in (("whole image", np.ones_like(hit, bool)), ("without sphere 11", hit != 11), ("sphere 11", hit == 11)):
        print(f"{name:18s} PSNR {psnr(col, ref, m):6.2f} dB   equal {np.mean(d[m] == 0):.3f}   "
              f"within 2 {np.mean(d[m] <= 2):.3f}   within 16 {np.mean(d[m] <= 16):.3f}")
    for k in (2, 4, 8):
        print(f"box {k}x: PSNR {psnr(T.box(col, k), T.box(ref, k)):.2f} dB")
    Image.fromarray(np.clip(d * 4, 0, 255).astype(np.uint8)).save(os.path.join(OUT, "thumbnail_diff.png"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "render":
        render(int(sys.argv[2]) if len(sys.argv) > 2 else 4096)
    else:
        compare(*sys.argv[2:3])
