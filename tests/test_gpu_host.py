"""GPU test of the C++ host layer (Application -> Scene -> Engine -> ComputePipeline -> C ABI):
the headless driver rt_headless runs a scripted interactive session -- key presses that move the
camera (accumulation resets while it moves), a mouse drag (reset + new basis), a window resize --
and its final image must equal, bit for bit, a Python re-enactment of the reference's main loop
(src/application.cpp:28-118, includes/scene.hpp:106-116) that renders with the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle as O
import raytracing_amd as rt
from conftest import ROOT

pytestmark = pytest.mark.gpu

BINARY = os.path.join(ROOT, "raytracing_amd", "rt_headless")


def _read_ppm(path):
    data = open(path, "rb").read()
    magic, dims, maxv, rest = data.split(b"\n", 3)
    assert magic == b"P6" and maxv == b"255"
    w, h = (int(v) for v in dims.split())
    return np.frombuffer(rest, np.uint8).reshape(h, w, 3)


def _simulate(script, frames, w, h, dt=np.float32(1.0 / 60.0)):
    """The reference's per-frame order: handleInput, Scene::update(dt), render."""
    scene = rt.Scene.load()
    ubo = scene.camera.copy()
    yaw, pitch = np.float32(90.0), np.float32(0.0)
    f, r, u = O.camera_basis(float(yaw), float(pitch))  # src/application.cpp:10-23
    ubo["camera_forward"][0], ubo["camera_right"][0], ubo["camera_up"][0] = f, r, u
    acc, vel = np.zeros(3, np.float32), np.zeros(3, np.float32)
    keys = set()
    accum = np.zeros((h, w, 4), np.float32)
    color = None
    counts = []
    spheres = scene.spheres.copy()
    for frame in range(frames):
        resized = None
        ui = []
        for ev in [e for e in script if e[0] == frame]:
            if ev[1] == "key":
                (keys.add if ev[3] == "down" else keys.discard)(ev[2])
            elif ev[1] == "mouse":
                yaw = np.float32(yaw + np.float32(ev[2]) * np.float32(0.25))
                pitch = np.float32(np.clip(pitch + np.float32(ev[3]) * np.float32(0.25), -89.0, 89.0))
                f, r, u = O.camera_basis(float(yaw), float(pitch))
                ubo["camera_forward"][0], ubo["camera_right"][0], ubo["camera_up"][0] = f, r, u
                ubo["frameCount"][0] = 0
            elif ev[1] == "resize":
                resized = (ev[2], ev[3])
            elif ev[1] == "ui":
                ui.append(ev[2:])
        speed = np.float32(100.0)
        for key, vec, sign in (("W", "camera_forward", 1), ("S", "camera_forward", -1), ("A", "camera_right", -1),
                               ("D", "camera_right", 1), ("E", "camera_up", -1), ("Q", "camera_up", 1)):
            if key in keys:
                acc += np.float32(sign) * (ubo[vec][0] * speed)
        O.scene_update(ubo, acc, vel, float(dt))
        counts.append(int(ubo["frameCount"][0]))
        accum, color, _ = O.render(ubo, spheres, w, h, n_frames=1, accum=accum, threads=8)
        # the overlay's widgets act after this frame's dispatch (src/engine/engine.cpp:67-68,
        # src/engine/src/graphics_pipeline.cpp:92-121)
        for act in ui:
            if act[0] == "reset":
                ubo["frameCount"][0] = 0
            elif act[0] == "camera":
                old = ubo["camera_position"][0].copy()
                for k, v in enumerate(act[1:4]):
                    if v != "=":
                        ubo["camera_position"][0][k] = np.clip(np.float32(v), -100.0, 100.0)
                if not np.array_equal(old, ubo["camera_position"][0]):
                    ubo["frameCount"][0] = 0
            elif act[0] == "sphere":
                for k, v in enumerate(act[2:5]):
                    if v != "=":
                        spheres["center"][act[1]][k] = np.clip(np.float32(v), -10.0, 10.0)
        if resized:  # Engine::render: the resize takes effect after this frame's dispatch
            w, h = resized
            accum = np.zeros((h, w, 4), np.float32)
            ubo["frameCount"][0] = 0
    return color, counts


def test_headless_session_matches_reenactment(tmp_path):
    assert os.path.exists(BINARY), "build rt_headless first (make -C raytracing_amd/csrc)"
    script = [(0, "key", "W", "down"), (2, "key", "D", "down"), (4, "key", "W", "up"), (5, "key", "D", "up"),
              (9, "mouse", 37.0, -11.0), (14, "resize", 200, 120), (15, "key", "Q", "down"), (16, "key", "Q", "up")]
    path = tmp_path / "session.txt"
    with open(path, "w") as f:
        f.write("# frame event\n")
        for ev in script:
            f.write(" ".join(str(x) for x in ev) + "\n")
    out = tmp_path / "out.ppm"
    env = dict(os.environ, RT_SCENE=rt.DEFAULT_SCENE)
    res = subprocess.run([BINARY, "--size", "256x144", "--frames", "24", "--script", str(path), "--out", str(out)],
                         env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    color, counts = _simulate(script, 24, 256, 144)
    got = _read_ppm(out)
    assert got.shape == (120, 200, 3)
    assert np.array_equal(got, color[..., :3])
    assert f"accumulated frame count {counts[-1]}" in res.stdout
    assert 1 in counts[1:9] and counts[-1] >= 1


def test_headless_overlay_widgets_act_one_frame_late(tmp_path):
    """The settings window's widgets as script events: the reset button and the camera sliders
    restart the accumulation, the sphere sliders move a sphere WITHOUT restarting it
    (src/engine/src/graphics_pipeline.cpp:92-121), and all of them act after the frame's dispatch."""
    script = [(2, "ui", "sphere", 5, 1.5, "=", -2.25), (4, "ui", "reset"), (6, "ui", "camera", "=", 7.5, "="),
              (7, "ui", "camera", "=", 7.5, "="),  # unchanged value: no reset
              (8, "ui", "sphere", 0, 250.0, "=", "="),  # slider range: clamps to 10
              (9, "ui", "camera", -500.0, "=", "=")]
    path = tmp_path / "ui.txt"
    with open(path, "w") as f:
        for ev in script:
            f.write(" ".join(str(x) for x in ev) + "\n")
    out = tmp_path / "out.ppm"
    env = dict(os.environ, RT_SCENE=rt.DEFAULT_SCENE)
    for frames in (4, 6, 12):  # before / after the reset, and the whole session
        res = subprocess.run([BINARY, "--size", "192x112", "--frames", str(frames), "--script", str(path), "--out", str(out)],
                             env=env, capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stdout + res.stderr
        color, counts = _simulate(script, frames, 192, 112)
        assert np.array_equal(_read_ppm(out), color[..., :3]), frames
        assert f"accumulated frame count {counts[-1]}" in res.stdout
    # frame counts seen by the kernel: the sphere edit at frame 2 does not restart them, the button
    # at frame 4 shows at frame 5, the camera slider at 6 shows at 7, the repeated value does not reset
    assert counts == [1, 2, 3, 4, 5, 1, 2, 1, 2, 3, 1, 2]


def test_headless_reports_errors_like_the_reference(tmp_path):
    """main.cpp:5-11: print the exception's message, exit code 1."""
    env = dict(os.environ, RT_SCENE=str(tmp_path / "missing.yaml"))
    res = subprocess.run([BINARY, "--frames", "1"], env=env, capture_output=True, text=True, timeout=60)
    assert res.returncode == 1 and "cannot open scene file" in res.stdout


def test_headless_renders_a_scene_file_with_materials(tmp_path):
    """The C++ host uploads the optional per-sphere materials of a scene file (4th sphere element):
    8 still frames of the RTIOW layout with materials equal the oracle's."""
    scene = rt.Scene.generate(rt.KIND_RTIOW_MATERIALS, 0, 5)
    path = tmp_path / "materials.yaml"
    scene.save(str(path))
    out = tmp_path / "out.ppm"
    res = subprocess.run([BINARY, "--size", "192x112", "--frames", "8", "--scene", str(path), "--out", str(out)],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    loaded = rt.Scene.load(str(path))
    ubo = loaded.camera.copy()
    # Application::Application installs the launch orientation (src/application.cpp:10-23)
    f, r, u = O.camera_basis(90.0, 0.0)
    ubo["camera_forward"][0], ubo["camera_right"][0], ubo["camera_up"][0] = f, r, u
    ubo["frameCount"][0] = 1
    _, color, _ = O.render(ubo, loaded.spheres, 192, 112, n_frames=8, materials=loaded.materials, threads=8)
    plain = O.render(ubo, loaded.spheres, 192, 112, n_frames=8, threads=8)[1]
    got = _read_ppm(out)
    assert np.array_equal(got, color[..., :3])
    assert not np.array_equal(got, plain[..., :3])


def test_headless_reproduces_the_reference_screenshot(tmp_path):
    """The C++ host, end to end: default scene file, the camera pose recovered for the reference's
    thumbnail.png, 300 frames at the screenshot's size -> the written image matches the screenshot
    (reduced 4x: PSNR >= 29.5 dB, 8x: >= 32.5 dB)."""
    import json

    import thumbnail as T
    from conftest import GOLDEN

    pose = json.load(open(os.path.join(GOLDEN, "thumbnail_pose.json")))
    out = tmp_path / "thumb.ppm"
    cmd = [BINARY, "--scene", rt.DEFAULT_SCENE, "--size", f"{T.WIDTH}x{T.HEIGHT}", "--frames", "300",
           "--position", ",".join(repr(v) for v in pose["position"]), "--yaw", repr(pose["yaw_degrees"]),
           "--pitch", repr(pose["pitch_degrees"]), "--out", str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    got = _read_ppm(out).astype(np.float64)[:1008, :1904]
    ref = T.reduced_screenshot()
    mine = T.box(got * (301.0 / 300.0), 4)   # the colour image divides by frameCount + 1
    assert T.psnr(mine, ref) >= 29.5 and T.psnr(T.box(mine, 2), T.box(ref, 2)) >= 32.5
