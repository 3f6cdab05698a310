"""Partitioning the render across GPUs (one process per GPU, torch.distributed).

The reference is single-GPU.  Pixels and frames are independent, so the path shards two ways,
each with one exchange step:

SampleSplit  rank g of G renders frame numbers g+1, g+1+G, ... (the RNG formula is untouched,
             so the union of samples is exactly the single-GPU sample set), accumulates locally,
             and one sum-reduce of the float4 accumulation image to rank 0 ends the job.  Only
             the float summation order differs from a single-GPU run.
TileSplit    rank g renders a band of 8x8-tile rows every frame (interactive mode) and sends its
             band of the rgba8 image to rank 0; bit-identical to a single-GPU frame.

Both classes only compute schedules and move tensors; the rendering is done by whatever
`renderer` object they are handed (raytracing_amd.Renderer on a GPU).
"""
from __future__ import annotations

from .renderer import CLEAR_FIRST, CLEAR_NEVER


def _dist():
    import torch.distributed as dist

    return dist


def _current_stream() -> int:
    """The raw hipStream_t of torch's current stream: collectives on device tensors are ordered
    against it (ProcessGroupNCCL makes its own stream wait for it, and Work.wait() makes it wait
    for the collective)."""
    import torch

    return torch.cuda.current_stream().cuda_stream


class SampleSplit:
    def __init__(self, rank: int, world: int):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        self.rank, self.world = rank, world

    def schedule(self, step: int, frames_per_step: int, restart: bool = False):
        """(first frameCount, frame stride, number of frames, clear mode) of this rank at `step`.

        Step s covers the global frames s*F*G+1 .. (s+1)*F*G; rank g takes every G-th of them
        starting at g+1.  The accumulation is zeroed at step 0 (frame 1 is where the reference
        clears, res/shaders/shader.comp:79-81; every rank has to clear its own copy)."""
        first = step * frames_per_step * self.world + self.rank + 1
        clear = CLEAR_FIRST if (step == 0 or restart) else CLEAR_NEVER
        return first, self.world, frames_per_step, clear

    def dispatch(self, renderer, scene, step: int, frames_per_step: int, restart: bool = False,
                 timed: bool = False) -> None:
        first, stride, n, clear = self.schedule(step, frames_per_step, restart)
        renderer.set_frame_schedule(stride, clear)
        ubo = scene.camera.copy()
        ubo["frameCount"][0] = first
        renderer.set_uniforms(ubo)
        if timed:
            renderer.dispatch_timed(n)  # with an event pair of its own (Renderer.timed_total)
        else:
            renderer.dispatch(n)

    def reduce(self, accum, dst: int = 0, renderer=None) -> None:
        """Sum the per-rank accumulation images into rank `dst` (RCCL on GPUs, gloo on CPU).

        With `renderer` (the context that renders into `accum`) the reduce is ordered against it on
        the device, both ways and without blocking the host: it starts when the dispatches issued
        so far are complete, and later dispatches -- which may clear or add to `accum` on any rank
        -- start when it has read the image (over RCCL a collective returns once it is queued).
        Without, the caller synchronises."""
        if self.world > 1:
            dist = _dist()
            ordered = renderer is not None and accum.is_cuda
            if ordered:
                renderer.stream_wait(_current_stream())
            if accum.is_cuda and dist.get_backend() == "gloo":  # rehearsal on one GPU: stage through the host
                host = accum.cpu()
                dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
                if self.rank == dst:
                    accum.copy_(host)
            else:
                dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
            if ordered:
                renderer.wait_stream(_current_stream())

    def divisor(self, steps: int, frames_per_step: int) -> float:
        """The shader divides by frameCount + 1 (shader.comp:129); frameCount of the last frame
        of the whole job is the total number of frames."""
        return float(steps * frames_per_step * self.world + 1)


class TileSplit:
    def __init__(self, rank: int, world: int, height: int):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        self.rank, self.world = rank, world
        self.tile_rows = height // 8  # the reference dispatches floor(H/8) rows of groups
        base, extra = divmod(self.tile_rows, world)
        starts = [g * base + min(g, extra) for g in range(world + 1)]
        self.bands = [(starts[g], starts[g + 1]) for g in range(world)]

    @property
    def rows(self):
        return self.bands[self.rank]

    def apply(self, renderer) -> None:
        r0, r1 = self.rows
        if r1 <= r0:
            # a rank with no rows renders nothing: an empty band is expressed as [rows, rows)
            renderer.set_tile_rows(self.tile_rows, self.tile_rows)
        else:
            renderer.set_tile_rows(r0, r1)

    def gather(self, image, dst: int = 0) -> None:
        """Move every rank's band of `image` (H x W [x C], rows = pixels) into rank `dst`'s copy."""
        if self.world == 1:
            return
        dist = _dist()
        # Both sides use the batched primitive (over RCCL it runs on the world communicator; a plain
        # send would lazily create a two-rank communicator that the batched receive does not use):
        # all receives of a frame are one grouped launch on the display rank, one send on the others.
        if self.rank == dst:
            ops = [dist.P2POp(dist.irecv, image[r0 * 8:r1 * 8], g)
                   for g, (r0, r1) in enumerate(self.bands) if g != dst and r1 > r0]
        else:
            r0, r1 = self.rows
            ops = [dist.P2POp(dist.isend, image[r0 * 8:r1 * 8], dst)] if r1 > r0 else []
        for q in dist.batch_isend_irecv(ops) if ops else []:
            q.wait()


class InteractiveLoop:
    """The interactive tile split with two frames in flight, as the reference keeps them
    (src/engine/includes/config.hpp:12): frame f is traced into one of two colour images while the
    bands of frame f-1 travel to the display rank out of the other, so the exchange of a frame
    overlaps the tracing of the next one, and frame f is dispatched BEFORE the host waits for frame
    f-1, so the GPU goes from one frame to the next without waiting for the host.

    `renderer` needs set_uniforms / dispatch / sync / sync_dispatch; `bind_color(k)` makes colour
    image k (0 or 1) the one the NEXT dispatch writes, without waiting (Renderer.bind_buffers_next);
    `image_of(k)` is the H x W [x C] tensor of image k that is exchanged (a host copy when the
    backend cannot move device memory).  `on_frame(f, image)` is called on the display rank with
    every completed frame, in order.

    `transfer_stream()` returns the raw HIP stream that the exchange of device images is queued
    on (torch's current stream for RCCL), or is None when the exchange blocks the host (gloo, host
    copies).  Over RCCL a send returns as soon as it is QUEUED, so the image it reads must not be
    handed to the next dispatch by the host's say-so alone: after every exchange the renderer's
    stream is made to wait for the transfer stream (Renderer.wait_stream), which is the fence of
    the reference's frame slot (src/engine/engine.cpp:48,132) and costs the host nothing.
    """

    def __init__(self, tiles: TileSplit, renderer, bind_color, image_of, on_frame=None, dst: int = 0,
                 dispatch=None, transfer_stream=None):
        self.tiles, self.renderer, self.bind_color, self.image_of = tiles, renderer, bind_color, image_of
        self.on_frame, self.dst = on_frame, dst
        self.dispatch = dispatch or (lambda: renderer.dispatch(1))
        self.transfer_stream = transfer_stream
        self.count = 0

    def _deliver(self, f: int) -> None:
        image = self.image_of(f % 2)
        self.tiles.gather(image, dst=self.dst)
        if self.transfer_stream is not None:
            # image f % 2 is written again by frame f + 2: not before this transfer has read it
            self.renderer.wait_stream(self.transfer_stream())
        if self.on_frame is not None and self.tiles.rank == self.dst:
            self.on_frame(f, image)

    def frame(self, ubo) -> None:
        f = self.count
        self.bind_color(f % 2)          # frame f-2, the last to write this image: its transfer is waited for on the device
        self.renderer.set_uniforms(ubo)
        self.dispatch()                 # asynchronous on a GPU, queued behind frame f-1
        if f > 0:
            self.renderer.sync_dispatch(1)  # frame f-1 is complete; frame f is being traced
            self._deliver(f - 1)
        self.count = f + 1

    def finish(self) -> None:
        if self.count > 0:
            self.renderer.sync()
            self._deliver(self.count - 1)
            self.count = 0


class SharedFrameFlags:
    """Frame counters that the processes of one node share through POSIX shared memory: slot g
    holds the number of frames whose band rank g's HOST has seen complete.  Plain host stores and
    loads -- no collective, no device-side spinning; a wait that lasts longer than `timeout`
    seconds raises instead of hanging."""

    def __init__(self, world: int, name: str | None = None, timeout: float = 30.0):
        from multiprocessing import shared_memory

        import numpy as np

        self.world, self.timeout = world, timeout
        self.owner = name is None
        self.shm = shared_memory.SharedMemory(create=True, size=64 * world) if self.owner \
            else shared_memory.SharedMemory(name=name)
        # one counter per cache line
        self.done = np.ndarray((world, 8), dtype=np.int64, buffer=self.shm.buf)[:, 0]
        if self.owner:
            self.done[:] = 0

    @property
    def name(self) -> str:
        return self.shm.name

    def publish(self, rank: int, frames_done: int) -> None:
        self.done[rank] = frames_done

    def wait_all(self, frames_done: int, ranks) -> None:
        import time

        t0 = None
        for g in ranks:
            while int(self.done[g]) < frames_done:
                if t0 is None:
                    t0 = time.monotonic()
                elif time.monotonic() - t0 > self.timeout:
                    raise TimeoutError(f"rank {g} has not finished frame {frames_done} after {self.timeout} s")

    def close(self) -> None:
        self.done = None
        self.shm.close()
        if self.owner:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass


class PeerImages:
    """The display rank's two colour images, mapped into every rank of the node (rt_ipc_alloc /
    rt_ipc_open): each rank's kernel writes its band of rows straight into them over xGMI."""

    def __init__(self, rank: int, dst: int, device: int, nbytes: int, handles=None):
        import ctypes as C

        from . import _native as N

        self.rank, self.dst, self.device, self.nbytes = rank, dst, device, nbytes
        self.owner = rank == dst
        self.ptrs, self.handles = [], []
        for k in range(2):
            p = C.c_void_p()
            if self.owner:
                h = C.create_string_buffer(64)
                N.check(N.lib().rt_ipc_alloc(device, nbytes, C.byref(p), h))
                self.handles.append(h.raw)
            else:
                N.check(N.lib().rt_ipc_open(device, handles[k], C.byref(p)))
            self.ptrs.append(p.value)

    def close(self) -> None:
        from . import _native as N

        for p in self.ptrs:
            N.check((N.lib().rt_ipc_free if self.owner else N.lib().rt_ipc_close)(self.device, p))
        self.ptrs = []


class DirectTileLoop:
    """The interactive tile split without a transfer: every rank's kernel writes its band into the
    display rank's colour image (PeerImages), and what is exchanged per frame is a counter in
    shared host memory (SharedFrameFlags).  Two frames in flight as in InteractiveLoop: frame f is
    dispatched before the host waits for frame f-1.

    Order of events that makes the displayed frame whole and its own:
      * rank g publishes "f frames done" only after ITS HOST has seen the dispatch of frame f-1
        complete (Renderer.sync_dispatch), which is when that kernel's writes are in the display
        GPU's memory;
      * the display rank shows frame f-1 once every rank with a band has published it, and then
        publishes "f frames shown";
      * image f % 2 is written again by frame f + 2, which a rank dispatches only after the
        display rank has shown frame f (`shown` counter), so a frame on display is never written.
    `on_frame(f, k)` is called on the display rank, in order, with the frame number and the image
    index; it must have finished READING image k when it returns.
    """

    def __init__(self, tiles: TileSplit, renderer, bind_color, flags: SharedFrameFlags, shown: SharedFrameFlags,
                 on_frame=None, dst: int = 0, dispatch=None):
        self.tiles, self.renderer, self.bind_color = tiles, renderer, bind_color
        self.flags, self.shown, self.on_frame, self.dst = flags, shown, on_frame, dst
        self.dispatch = dispatch or (lambda: renderer.dispatch(1))
        self.writers = [g for g, (r0, r1) in enumerate(tiles.bands) if r1 > r0]
        self.count = 0

    def _finished(self, f: int) -> None:
        """Frame f (0-based) is complete on this rank's GPU."""
        self.flags.publish(self.tiles.rank, f + 1)
        if self.tiles.rank == self.dst:
            self.flags.wait_all(f + 1, self.writers)
            if self.on_frame is not None:
                self.on_frame(f, f % 2)
            self.shown.publish(0, f + 1)

    def frame(self, ubo) -> None:
        f = self.count
        if f >= 2:
            self.shown.wait_all(f - 1, [0])  # frame f-2, the last to use image f % 2, has been shown
        self.bind_color(f % 2)
        self.renderer.set_uniforms(ubo)
        self.dispatch()
        if f > 0:
            self.renderer.sync_dispatch(1)  # frame f-1 is complete; frame f is being traced
            self._finished(f - 1)
        self.count = f + 1

    def finish(self) -> None:
        if self.count > 0:
            self.renderer.sync()
            self._finished(self.count - 1)
            self.count = 0
