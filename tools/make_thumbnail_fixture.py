#!/usr/bin/env python3
"""Derive tests/golden/thumbnail_sky.json from the one rendered output the reference ships,
thumbnail.png (a screenshot at an unrecorded camera pose): its most frequent colour is the
converged sky pixel, which does not depend on the pose; the next two are sphere surfaces that
mirror the sky (one bounce), which depend only on the sphere's albedo.

It also writes tests/golden/thumbnail_x4.npz, the screenshot reduced 4x with a box filter, which
the pose-based comparison (tests/golden/thumbnail_pose.json) is checked against.

usage: python tools/make_thumbnail_fixture.py [/root/reference/thumbnail.png]
"""
import collections
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/thumbnail.png"
img = np.array(Image.open(path).convert("RGB"))
top = collections.Counter(map(tuple, img.reshape(-1, 3))).most_common(3)
(rgb, count) = top[0]
out = {
    "source": "thumbnail.png of the reference: its three most frequent colours -- the sky after many accumulated "
              "frames, and two large flat-coloured areas (sphere surfaces that mirror the sky)",
    "image_size": [int(img.shape[1]), int(img.shape[0])],
    "sky_rgb": [int(v) for v in rgb],
    "pixels_with_that_colour": int(count),
    "surface_rgb": [{"rgb": [int(v) for v in c], "pixels": int(n)} for c, n in top[1:]],
    # The 14 most frequent colours.  Apart from the two saturated emitters they are flat areas whose
    # paths end in the sky after mirroring off one or two spheres: sky * 0.15 * albedo_a (* albedo_b).
    # With the albedos of the reference's scene file every one of them comes out exactly for any
    # accumulated frame count from 171 to 416 (the /(frameCount+1) of the resolve darkens slightly).
    "frequent_rgb": [{"rgb": [int(v) for v in c], "pixels": int(n)}
                     for c, n in collections.Counter(map(tuple, img.reshape(-1, 3))).most_common(14)],
    "frame_count_range": [171, 416],
}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "thumbnail_sky.json"), "w"), indent=1)
# the screenshot itself, 4x box-filtered (its top-left 1904x1008, the part an 8x8-tiled dispatch covers)
small = Image.open(path).convert("RGB").crop((0, 0, 1904, 1008)).resize((476, 252), Image.BOX)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "thumbnail_x4.npz"), rgb=np.array(small))
print(out)
