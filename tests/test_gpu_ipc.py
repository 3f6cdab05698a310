"""GPU test of the screen-tile split without a transfer: two PROCESSES on one GPU (inter-process
memory handles work on the same device as across xGMI peers), each tracing its band of tile rows
straight into the display process's colour images (rt_ipc_alloc / rt_ipc_open), frame counters in
shared host memory in place of a per-frame collective (raytracing_amd.multi_gpu.DirectTileLoop).
Every displayed frame must be the single-process frame, bit for bit and in order."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

W, H = 320, 200  # 25 tile rows: bands of 13 and 12
N_FRAMES = 12


def _ubos(n):
    import raytracing_amd as rt

    scene = rt.Scene.default()
    out, count = [], 0
    for i in range(n):
        moved = i % 3 != 2  # two frames in three restart the accumulation from a camera that has moved
        if moved:
            scene.camera["camera_position"][0][0] += 0.6
        count = 1 if moved else count + 1
        ubo = scene.camera.copy()
        ubo["frameCount"][0] = count
        out.append(ubo)
    return scene, out


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    import raytracing_amd as rt
    from raytracing_amd.multi_gpu import DirectTileLoop, PeerImages, SharedFrameFlags, TileSplit

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scene, ubos = _ubos(N_FRAMES)
        tiles = TileSplit(rank, world, H)
        box = [None]
        flags = shown = images = None
        if rank == 0:
            flags, shown = SharedFrameFlags(world), SharedFrameFlags(1)
            images = PeerImages(rank, 0, 0, W * H * 4)
            box[0] = (images.handles, flags.name, shown.name)
        dist.broadcast_object_list(box, src=0)
        if rank != 0:
            handles, fname, sname = box[0]
            flags, shown = SharedFrameFlags(world, fname), SharedFrameFlags(1, sname)
            images = PeerImages(rank, 0, 0, W * H * 4, handles)
        import ctypes as C

        from test_zz_no_alloc import _Hip

        hip = _Hip()
        frames = []

        def on_frame(f, k):  # the display: a copy of image k, finished before it returns
            frames.append(hip.download(C.c_void_p(images.ptrs[k]), np.empty((H, W), np.uint32)))

        with rt.Renderer(W, H) as r:
            r.upload_spheres(scene.spheres)
            tiles.apply(r)
            # the accumulation image stays the context's own (a band of it per process); the colour image is the shared one
            loop = DirectTileLoop(tiles, r, lambda k: r.bind_buffers_next(0, images.ptrs[k]), flags, shown,
                                  on_frame=on_frame if rank == 0 else None)
            for ubo in ubos:
                loop.frame(ubo)
            loop.finish()
            r.bind_buffers(None, None)
        dist.barrier()  # nobody unmaps or frees while another rank may still write
        if rank == 0:
            np.save(os.path.join(out_dir, "frames.npy"), np.stack(frames))
        if rank != 0:
            images.close()
        dist.barrier()
        if rank == 0:
            images.close()
        flags.close()
        shown.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_processes_write_one_image(tmp_path):
    import raytracing_amd as rt

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(rank), "2", str(port), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for rank in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    got = np.load(tmp_path / "frames.npy")
    scene, ubos = _ubos(N_FRAMES)
    assert got.shape[0] == N_FRAMES
    with rt.Renderer(W, H) as r:
        r.upload_spheres(scene.spheres)
        for f, ubo in enumerate(ubos):
            r.set_uniforms(ubo)
            r.dispatch(1)
            want = r.read_color().view(np.uint32)[..., 0]
            assert np.array_equal(got[f], want), f"frame {f}"


if __name__ == "__main__":
    _worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
