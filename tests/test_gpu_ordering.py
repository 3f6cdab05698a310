"""GPU tests of the device-side ordering between the renderer's stream and a transfer stream
(rt_wait_stream / rt_stream_wait): what keeps the interactive tile split from handing an image to
the next frame while the exchange of the frame before still reads it, and a sum-reduce from
reading an accumulation image that is not complete.

One GPU stands in for the node: the exchange of a band is an ASYNCHRONOUS device copy on a second
stream, slowed by a large dummy copy queued in front of it, which is what a send over RCCL looks
like to the host (it returns once it is queued).  The reference keeps two frames in flight behind
a fence per frame slot (src/engine/engine.cpp:48,132, src/engine/includes/config.hpp:12)."""
import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd.multi_gpu import InteractiveLoop, TileSplit

pytestmark = pytest.mark.gpu

W, H = 640, 360
N_FRAMES = 9


def _moving_ubos(n):
    """Every frame restarts the accumulation from a camera that has moved: consecutive frames differ
    in most pixels."""
    scene = rt.Scene.default()
    out = []
    for i in range(n):
        ubo = scene.camera.copy()
        ubo["camera_position"][0][0] += 0.75 * i
        ubo["camera_position"][0][1] += 0.25 * i
        ubo["frameCount"][0] = 1
        out.append(ubo)
    return scene, out


def _expected(scene, ubos):
    frames = []
    with rt.Renderer(W, H) as r:
        r.upload_spheres(scene.spheres)
        for ubo in ubos:
            r.set_uniforms(ubo)
            r.dispatch(1)
            frames.append(r.read_color().view(np.int32)[..., 0].copy())
    return frames


class _SlowExchange:
    """TileSplit.gather for a world of one: the "transfer" of an image is a device copy on a side
    stream behind ~1.5 GiB of dummy copies, recorded frame by frame."""

    rank, bands, rows = 0, [(0, H // 8)], (0, H // 8)

    def __init__(self, torch):
        self.torch = torch
        self.stream = torch.cuda.Stream()
        self.ballast = torch.empty(384 << 20, dtype=torch.uint8, device="cuda")
        self.sink = torch.empty_like(self.ballast)
        self.delivered = []

    def gather(self, image, dst=0):
        torch = self.torch
        with torch.cuda.stream(self.stream):
            for _ in range(4):
                self.sink.copy_(self.ballast, non_blocking=True)
            out = torch.empty_like(image)
            out.copy_(image, non_blocking=True)
        self.delivered.append(out)


def _run_pipelined(ordered):
    torch = pytest.importorskip("torch")
    scene, ubos = _moving_ubos(N_FRAMES)
    colors = [torch.zeros((H, W), dtype=torch.int32, device="cuda") for _ in range(2)]
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    exchange = _SlowExchange(torch)
    with rt.Renderer(W, H) as r:
        r.upload_spheres(scene.spheres)
        loop = InteractiveLoop(exchange, r, lambda k: r.bind_buffers_next(accum.data_ptr(), colors[k].data_ptr()),
                               lambda k: colors[k],
                               transfer_stream=(lambda: exchange.stream.cuda_stream) if ordered else None)
        for ubo in ubos:
            loop.frame(ubo)
        loop.finish()
        torch.cuda.synchronize()
    return scene, ubos, [d.cpu().numpy() for d in exchange.delivered]


def test_delivered_frames_are_their_own_bits_when_the_exchange_is_asynchronous():
    scene, ubos, got = _run_pipelined(ordered=True)
    want = _expected(scene, ubos)
    assert len(got) == N_FRAMES
    for f in range(N_FRAMES):
        assert np.array_equal(got[f], want[f]), f"frame {f}: the delivered image is not frame {f}'s"
    # the frames do differ from one another, so a frame delivered out of the wrong dispatch would show
    assert all((want[f] != want[f + 2]).mean() > 0.02 for f in range(N_FRAMES - 2))


def test_without_the_ordering_the_slow_exchange_reads_a_later_frame():
    """The control: with no wait on the transfer stream the host hands image f % 2 to frame f + 2
    while the slow copy of frame f has not read it yet.  Reported, not asserted as a failure of the
    library: whether the race shows depends on the two streams really running side by side."""
    scene, ubos, got = _run_pipelined(ordered=False)
    want = _expected(scene, ubos)
    torn = [f for f in range(N_FRAMES) if not np.array_equal(got[f], want[f])]
    print(f"unordered control: {len(torn)} of {N_FRAMES} delivered frames are not their own: {torn}")
    if not torn:
        pytest.skip("the unordered exchange happened to finish in time on this machine")


def test_stream_wait_orders_a_reader_after_the_dispatch():
    """rt_stream_wait: a copy queued on another stream right after a dispatch, with no host
    synchronisation in between, reads the finished image."""
    torch = pytest.importorskip("torch")
    scene = rt.Scene.generate(rt.KIND_RTIOW, 0, 7)
    ubo = scene.camera.copy()
    ubo["frameCount"][0] = 1
    side = torch.cuda.Stream()
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    color = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    with rt.Renderer(W, H) as r:
        r.set_scene(scene, frame_count=1)
        r.dispatch(4)
        want_acc, want_col = r.read_accum(), r.read_color().view(np.int32)[..., 0]
        r.bind_buffers(accum.data_ptr(), color.data_ptr())
        torch.cuda.synchronize()
        r.dispatch(4)                      # a few hundred microseconds of tracing ...
        r.stream_wait(side.cuda_stream)    # ... that the copies below must not overtake
        with torch.cuda.stream(side):
            got_acc = accum.clone()
            got_col = color.clone()
        side.synchronize()
        assert np.array_equal(got_acc.cpu().numpy().view(np.uint32), want_acc.view(np.uint32))
        assert np.array_equal(got_col.cpu().numpy(), want_col)
        r.sync()
