"""World-size-2 tests of the multi-GPU host logic on CPU (gloo).

raytracing_amd.multi_gpu only computes frame/row schedules and moves tensors; the rendering
object it drives is a parameter.  Here that object is a stand-in backed by the CPU oracle, so
the partition + exchange can be checked against a single-process render without a GPU:
  SampleSplit -- frames split by stride + sum-reduce == the single-process sum of the same
                 frames (same per-frame values; only the summation order may differ);
  TileSplit   -- bands of tile rows + gather == the single-process image, bit for bit.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O
import raytracing_amd as rt
from raytracing_amd.multi_gpu import InteractiveLoop, SampleSplit, TileSplit

W, H = 96, 72  # 9 tile rows: an odd number, so the two bands differ in size


class OracleRenderer:
    """Stand-in with the Renderer methods multi_gpu uses, computing with the CPU oracle."""

    def __init__(self, scene, width, height):
        self.scene, self.w, self.h = scene, width, height
        self.accum = np.zeros((height, width, 4), np.float32)
        self.color = np.zeros((height, width, 4), np.uint8)
        self.stride, self.clear, self.rows, self.ubo = 1, 0, None, None

    def sync(self):
        pass

    def sync_dispatch(self, back):
        pass  # the stand-in's dispatch is synchronous

    def set_frame_schedule(self, stride, clear):
        self.stride, self.clear = stride, clear

    def set_tile_rows(self, r0, r1):
        self.rows = (r0, r1)

    def set_uniforms(self, ubo):
        self.ubo = ubo.copy()

    def dispatch(self, n):
        if self.rows is not None and self.rows[0] >= self.rows[1]:
            return
        self.accum, col, _ = O.render(self.ubo, self.scene.spheres, self.w, self.h, n_frames=n, frame_stride=self.stride,
                                      clear_mode=self.clear, rows=self.rows, accum=self.accum)
        rows = slice(None) if self.rows is None else slice(self.rows[0] * 8, self.rows[1] * 8)
        self.color[rows] = col[rows]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scene = rt.Scene.default()
        # --- sample split: 3 steps of 2 frames per rank = frames 1..12 over 2 ranks
        split = SampleSplit(rank, world)
        r = OracleRenderer(scene, W, H)
        for step in range(3):
            split.dispatch(r, scene, step=step, frames_per_step=2)
        acc = torch.from_numpy(r.accum)
        split.reduce(acc, dst=0)
        if rank == 0:
            np.save(os.path.join(out_dir, "sample_accum.npy"), acc.numpy())
            np.save(os.path.join(out_dir, "sample_divisor.npy"), np.array(split.divisor(3, 2)))
        # --- tile split: one frame, bands gathered to rank 0
        tiles = TileSplit(rank, world, H)
        r2 = OracleRenderer(scene, W, H)
        tiles.apply(r2)
        ubo = scene.camera.copy()
        ubo["frameCount"][0] = 1
        r2.set_uniforms(ubo)
        r2.dispatch(1)
        col = torch.from_numpy(r2.color)
        tiles.gather(col, dst=0)
        if rank == 0:
            np.save(os.path.join(out_dir, "tile_color.npy"), col.numpy())
            np.save(os.path.join(out_dir, "bands.npy"), np.array(tiles.bands))
        # --- interactive loop: 7 frames with a moving camera and resets, two colour images,
        #     frame f exchanged after frame f+1 has been dispatched
        r3 = OracleRenderer(scene, W, H)
        tiles.apply(r3)
        images = [torch.zeros((H, W, 4), dtype=torch.uint8) for _ in range(2)]
        shown = []

        def bind_color(k):
            r3.color = images[k].numpy()  # the stand-in writes straight into image k

        loop = InteractiveLoop(tiles, r3, bind_color, lambda k: images[k],
                               on_frame=lambda f, img: shown.append(img.numpy().copy()))
        for ubo in _interactive_ubos(7):
            loop.frame(ubo)
        loop.finish()
        if rank == 0:
            np.save(os.path.join(out_dir, "interactive.npy"), np.stack(shown))
    finally:
        dist.destroy_process_group()


def _interactive_ubos(n):
    """Uniform blocks of n frames in which the camera moves on some frames (frameCount restarts at
    1, as Scene::update does when the camera has moved) and rests on the others."""
    scene = rt.Scene.default()
    out, count = [], 0
    for i in range(n):
        moved = i in (0, 3, 5, 6)
        if moved:
            scene.camera["camera_position"][0][2] += 0.25
        count = 1 if moved else count + 1
        ubo = scene.camera.copy()
        ubo["frameCount"][0] = count
        out.append(ubo)
    return out


def test_schedules():
    s0, s1 = SampleSplit(0, 2), SampleSplit(1, 2)
    assert s0.schedule(0, 4) == (1, 2, 4, rt.CLEAR_FIRST)      # frames 1,3,5,7
    assert s1.schedule(0, 4) == (2, 2, 4, rt.CLEAR_FIRST)      # frames 2,4,6,8
    assert s0.schedule(1, 4) == (9, 2, 4, rt.CLEAR_NEVER)      # frames 9,11,13,15
    assert s1.schedule(2, 4)[0] == 18
    assert SampleSplit(0, 1).schedule(0, 8) == (1, 1, 8, rt.CLEAR_FIRST)
    assert s0.divisor(3, 4) == 25.0  # 24 frames rendered, the shader divides by frameCount + 1
    frames = sorted(f for g in range(8) for st in range(3)
                    for f in [SampleSplit(g, 8).schedule(st, 5)[0] + 8 * j for j in range(5)])
    assert frames == list(range(1, 121))  # every frame exactly once
    with pytest.raises(ValueError):
        SampleSplit(2, 2)
    bands = TileSplit(0, 8, 1080).bands  # 135 tile rows over 8 ranks
    assert bands[0] == (0, 17) and bands[-1][1] == 135 and sum(b - a for a, b in bands) == 135
    assert all(b - a in (16, 17) for a, b in bands)
    assert TileSplit(3, 4, 16).bands == [(0, 1), (1, 2), (2, 2), (2, 2)]  # more ranks than rows


def test_world_size_two_gloo(tmp_path, default_scene):
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    ubo = default_scene.camera.copy()
    ubo["frameCount"][0] = 1
    # sample split: rank 0 summed frames {1,3,5,7,9,11}, rank 1 {2,4,...,12}, then reduced
    a0, _, _ = O.render(ubo, default_scene.spheres, W, H, n_frames=6, frame_stride=2, clear_mode=O.CLEAR_FIRST)
    ubo["frameCount"][0] = 2
    a1, _, _ = O.render(ubo, default_scene.spheres, W, H, n_frames=6, frame_stride=2, clear_mode=O.CLEAR_FIRST)
    got = np.load(tmp_path / "sample_accum.npy")
    assert np.array_equal((a0 + a1).view(np.uint32), got.view(np.uint32))
    ubo["frameCount"][0] = 1
    full, _, _ = O.render(ubo, default_scene.spheres, W, H, n_frames=12)
    np.testing.assert_allclose(got, full, rtol=2e-6, atol=1e-6)  # summation order only
    assert np.all(got[:72, :, 3] == 12.0)
    assert float(np.load(tmp_path / "sample_divisor.npy")) == 13.0
    # tile split: bit-identical image
    _, col, _ = O.render(ubo, default_scene.spheres, W, H, n_frames=1)
    assert np.array_equal(np.load(tmp_path / "tile_color.npy"), col)
    assert np.load(tmp_path / "bands.npy").tolist() == [[0, 5], [5, 9]]
    # interactive loop: every displayed frame equals the single-process frame, bit for bit, in order
    shown = np.load(tmp_path / "interactive.npy")
    ubos = _interactive_ubos(7)
    assert shown.shape[0] == 7 and len({int(u["frameCount"][0]) for u in ubos}) > 1
    acc = None
    for f, u in enumerate(ubos):
        acc, col, _ = O.render(u, default_scene.spheres, W, H, n_frames=1, accum=acc)
        assert np.array_equal(shown[f], col), f


def test_interactive_camera_script_keeps_the_scene_in_view():
    """bench.py's interactive workload taps the view key, brakes and releases, forward and backward in
    turn: the camera must stay within two units of where it started (the default scene is 36 units
    ahead), and every 40-frame period must both restart the accumulation (frameCount == 1 while the
    camera moves more than 0.01 a frame, includes/scene.hpp:106-116) and add to it."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import raytracing_amd as rt

    start = rt.Scene.default().camera["camera_position"][0].copy()
    _, ubos = bench.interactive_ubos(rt, 440)
    pos = np.array([u["camera_position"][0] for u in ubos])
    count = np.array([int(u["frameCount"][0]) for u in ubos])
    assert np.abs(pos - start).max() < 2.0
    for p0 in range(0, 440, 40):
        period = count[p0:p0 + 40]
        assert 8 <= (period == 1).sum() <= 24 and period.max() >= 16, period
