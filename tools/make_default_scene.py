#!/usr/bin/env python3
"""Writes raytracing_amd/scenes/default40.yaml from the reference's scene file.

The reference's res/scenes/scene.yaml stores 1400 spheres but its camera row sets
sphereCount = 40 and nothing raises it, so only the first 40 are ever traced
(res/shaders/shader.comp:42).  This script keeps those 40 sphere entries and the
camera block, byte for byte, and drops the 1360 untraced ones.  It is data (an
input fixture), not code.

usage: tools/make_default_scene.py [/root/reference/res/scenes/scene.yaml]
"""
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/res/scenes/scene.yaml"
lines = open(src, "rb").read().split(b"\n")
assert lines[0] == b"spheres:"
cam = lines.index(b"camera:")
assert (cam - 1) % 10 == 0, "each sphere entry is 10 lines"
keep = lines[: 1 + 40 * 10] + lines[cam:]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracing_amd", "scenes", "default40.yaml")
open(out, "wb").write(b"\n".join(keep))
print("wrote", os.path.normpath(out), len(keep), "lines")
