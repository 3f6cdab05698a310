#!/usr/bin/env python3
"""bench.py -- the benchmark contract of this repository.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Workloads (one per configuration the baseline names; the default is the one the headline metric
is quoted on):

  rtiow4k         RTIOW-layout scene (484 spheres), 3840x2160, depth 50, 8 frames (spp) per step
  default720p_d8  the reference's default scene (40 traced spheres), 1280x720, depth 8, 64 frames per step
  default1080p    the same scene, 1920x1080, depth 50, 64 frames per step (16 steps = 1024 spp)
  stress4096      4096 random spheres, 3840x2160, depth 50, 2 frames per step
  interactive     default scene, 1920x1080, ONE frame per step with scripted camera motion and
                  accumulation resets; value = ms per displayed frame

One "step" is one dispatch of the path-trace kernel over the whole image; one ray is one
Trace() call of the shader (primary or bounce segment), counted inside the kernel.

Multi-GPU (one process per GPU; `--gpus N` without a launcher spawns the N ranks itself, under
torch.distributed.run the ranks come from the environment).  The throughput workloads split
SAMPLES: rank g of G renders frames g+1, g+1+G, ... with the unchanged RNG formula, accumulates
locally, and one RCCL sum-reduce of the float4 accumulation image to rank 0 ends the job, inside
the timed region (weak scaling: every rank renders steps x frames-per-step frames).  The
interactive workload splits SCREEN TILES: every rank traces a band of tile rows of every frame
and rank 0 gathers the rgba8 bands, the exchange of a frame overlapping the tracing of the next
(strong scaling).

Output: one JSON line on rank 0.  `roofline` prices the algorithmic work of the path-trace kernel
(21 FP32 flop per ray-sphere test that is evaluated) against the FP32 peak and, beside it, the
matrix pipe (where the sieve's dot products run above 64 spheres), the vector ALU's busy share
and HBM; `cpu_baseline` is the CPU oracle (a port of the shader, not the reference binary) on the
host cores, on a bounded sample of the same workload.  A default run on one GPU (no --workload)
also carries, measured after the headline's timed region, a short pass of each of the other four
workloads (`configs`) and of the headline workload with the camera-ray shortcuts switched off
(`brute_force_only`: every Trace() call walks the whole sphere list).
"""
import copy
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_TEST = 21             # algorithmic FP32 flop per ray-sphere test (a hoisted)
PEAK_FP32_TFLOPS = 157.3       # MI355X FP32 vector peak
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak
MFMA_FLOP_PER_TEST = 128       # executed on the matrix cores per test: 4 MFMAs of 32x32x16 per 32x32 tests
PEAK_HBM_GBPS = 8000.0         # MI355X HBM3E peak
BYTES_PER_PIXEL_DISPATCH = 36  # float4 read + float4 write + rgba8 write
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc.json")

WORKLOADS = {
    "rtiow4k": dict(scene="rtiow", size=(3840, 2160), depth=50, fps=8, steps=8, warmup=2,
                    metric="Mrays/sec (whole node) @4K ~500 spheres"),
    "default720p_d8": dict(scene="default", size=(1280, 720), depth=8, fps=64, steps=8, warmup=2,
                           metric="Mrays/sec (whole node), default scene 1280x720, depth 8"),
    "default1080p": dict(scene="default", size=(1920, 1080), depth=50, fps=64, steps=16, warmup=2,
                         metric="Mrays/sec (whole node), default scene 1080p, 1024 spp"),
    "stress4096": dict(scene="cube4096", size=(3840, 2160), depth=50, fps=2, steps=6, warmup=2,
                       metric="Mrays/sec (whole node), 4096-sphere stress scene"),
    "interactive": dict(scene="default", size=(1920, 1080), depth=50, fps=1, steps=400, warmup=40,
                        metric="ms/frame 1080p interactive"),
}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None)
    p.add_argument("--warmup", type=int, default=None)
    p.add_argument("--workload", choices=list(WORKLOADS), default=None,
                   help="default: rtiow4k, followed on one GPU by short passes of the other workloads (`configs`)")
    p.add_argument("--no-shortcuts", action="store_true",
                   help="RT_OPTION_CAMERA_SHORTCUTS off: every Trace() call evaluates the whole sphere list")
    p.add_argument("--frames-per-step", type=int, default=None)
    p.add_argument("--profile", choices=["strict", "fused"], default="strict")
    p.add_argument("--kernel", choices=["stream", "valu", "tile", "culled", "bvh"], default="stream",
                   help="stream: the brute-force kernel the metrics are quoted on (sieve on the matrix cores from 65 to "
                        "4096 spheres); valu: the same with the sieve on the vector ALU; culled: over a spatially sorted "
                        "sphere list, skipping groups no ray of a wave can touch; bvh: that list walked as a hierarchy "
                        "by every lane on its own (culled and bvh: same image, not brute force)")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU baseline leg (0 = skip)")
    p.add_argument("--max-depth", type=int, default=None)
    p.add_argument("--materials", action="store_true",
                   help="rtiow4k with the material extension (lambertian/metal/dielectric per sphere) switched on; "
                        "the headline line is quoted without it (the reference has one surface response)")
    a = p.parse_args()
    a.side_configs = a.workload is None
    a.workload = a.workload or "rtiow4k"
    w = WORKLOADS[a.workload]
    a.steps = w["steps"] if a.steps is None else a.steps
    a.warmup = w["warmup"] if a.warmup is None else a.warmup
    a.frames_per_step = w["fps"] if a.frames_per_step is None else a.frames_per_step
    a.max_depth = w["depth"] if a.max_depth is None else a.max_depth
    if a.workload == "interactive":
        a.frames_per_step = 1
    return a


# ------------------------------------------------------------------ launching the ranks

def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as child processes (this
    process has made no GPU call yet) and return the worst exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    return max((abs(c) for c in codes), default=0)


# ------------------------------------------------------------------ scenes and the CPU leg

def make_scene(rt, name, materials=False):
    if name == "rtiow":
        return rt.Scene.generate(rt.KIND_RTIOW_MATERIALS if materials else rt.KIND_RTIOW, 0, 7), "rtiow-layout scene seed 7"
    if name == "default":
        return rt.Scene.default(), "default scene of the reference (40 traced spheres)"
    n = int(name.replace("cube", ""))
    return rt.Scene.generate(rt.KIND_CUBE, n, 3), f"stress scene seed 3 ({n} random spheres)"


def host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores))))


def lavapipe_probe():
    """The baseline of choice would be the reference shader on Mesa lavapipe (CPU Vulkan).  That
    needs a Vulkan loader, the lvp ICD and a GLSL compiler on this machine; say what is there."""
    import ctypes.util
    import glob
    import shutil

    loader = ctypes.util.find_library("vulkan")
    icd = glob.glob("/usr/share/vulkan/icd.d/lvp_icd*.json") + glob.glob("/etc/vulkan/icd.d/lvp_icd*.json")
    glslc = shutil.which("glslc") or shutil.which("glslangValidator")
    if loader and icd and glslc:
        return "lavapipe present but no Vulkan host harness is part of this build"
    missing = [n for n, ok in (("libvulkan", loader), ("lvp ICD", icd), ("glslc/glslangValidator", glslc)) if not ok]
    return "lavapipe baseline not possible here: " + ", ".join(missing) + " not found"


def cpu_baseline(scene, size, profile, max_depth, seconds, interactive_frames=None):
    """The oracle, multithreaded over tile rows, on whole frames of the same workload until
    `seconds` have passed (at least one frame).  interactive_frames: the uniform blocks of the
    interactive workload's first frames, rendered one frame per call as the GPU does."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O

    w, h = size
    cores = host_cores()
    mats = scene.materials
    mats = mats if len(mats) and mats[:, 0].any() else None
    ubo = scene.camera.copy()
    acc, rays, frames = None, 0, 0
    t0 = time.perf_counter()
    while True:
        if interactive_frames is not None:
            ubo = interactive_frames[frames]
        else:
            ubo["frameCount"][0] = frames + 1
        acc, _, r = O.render(ubo, scene.spheres, w, h, max_depth=max_depth, profile=profile, n_frames=1, threads=cores,
                             accum=acc, materials=mats)
        rays += r
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or (interactive_frames is not None and frames >= len(interactive_frames)):
            break
    sample = (f"{frames} whole frame(s) of the same {w}x{h} workload, depth {max_depth}, {rays} rays in {dt:.1f} s "
              f"(CPU oracle, {'fused' if profile else 'strict'} profile); " + lavapipe_probe())
    if interactive_frames is not None:
        return {"value": dt / frames * 1e3, "unit": "ms/frame", "cores": cores, "kind": "port", "sample": sample}
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port", "sample": sample}


# ------------------------------------------------------------------ roofline

def pmc_record(key):
    """Counter-derived figures of this workload from the committed rocprofv3 passes (they cannot be
    collected inside a timed run): HBM traffic per launch, VALU busy share, clock."""
    try:
        return json.load(open(PMC_FILE)).get(key)
    except Exception:
        return None


def roofline(args, tests, launches, kernel_ms, pixels_per_launch, pmc_key, sieve_mode, listed_share=0.0):
    """Algorithmic FP32 work and HBM bytes of the average launch against the two roofs.
    tests = rays x sphereCount; listed_share = the share of the rays that were camera rays traced against
    the candidate list of their tile instead of the sphere list: their tests were never evaluated and are
    left out of `achieved`."""
    listed_tests = tests * listed_share
    tests = tests - listed_tests
    launch_ms = kernel_ms / max(1, launches)
    achieved = tests / max(1, launches) * FLOP_PER_TEST / (launch_ms * 1e-3) / 1e12
    hbm = pixels_per_launch * BYTES_PER_PIXEL_DISPATCH / (launch_ms * 1e-3) / 1e9
    pmc = pmc_record(pmc_key) or {}
    out = {
        "bound": "valu",
        "bound_detail": "FP32 vector ALU: the path has no matrix contraction, so MFMA does not bound it (the FP32 "
                        "vector peak equals the dense FP32 MFMA peak, 157.3 TFLOP/s); the HBM roof is reported beside it",
        "achieved": achieved,
        "peak": PEAK_FP32_TFLOPS,
        "unit": "TFLOP/s",
        "frac": achieved / PEAK_FP32_TFLOPS,
        "flop_per_test": FLOP_PER_TEST,
        "tests_not_evaluated": {"share_of_rays_x_spheres": listed_share,
                                "why": "a pixel's camera ray is the same ray in every frame: its hit is found once per camera "
                                       "and scene (cone test per 8x8 tile, literal test of the few spheres left) and reused; "
                                       "`achieved` counts the other rays' tests only"},
        "kernel_ms_per_launch": launch_ms,
        "launches": launches,
        "hbm": {"achieved": hbm, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": hbm / PEAK_HBM_GBPS,
                "bytes_per_pixel_per_launch": BYTES_PER_PIXEL_DISPATCH},
        "traffic": pmc.get("traffic_bytes_per_launch"),
        "traffic_source": (f"profiles/{os.path.basename(PMC_FILE)}[{pmc_key}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                           "passes of this command, FETCH_SIZE doubled (gfx950), collected separately, not in this run")
                          if pmc.get("traffic_bytes_per_launch") is not None else None,
    }
    for k in ("valu_busy", "mfma_busy", "executed_flop_frac", "clock_ghz", "valu_insts_per_launch"):
        if k in pmc:
            out[k] = pmc[k]
    if sieve_mode == 3:
        executed = tests / max(1, launches) * MFMA_FLOP_PER_TEST / (launch_ms * 1e-3) / 1e12
        out["bound"] = "mfma"
        out["bound_detail"] = ("the sieve's two K=4 dot products (rays x spheres) run on the matrix cores as exact three-term "
                               "bf16 splits (v_mfma_f32_32x32x16_bf16, 128 executed flop per test), the vector ALU keeps 2 "
                               "instructions per test; `achieved`/`frac` stay the ALGORITHMIC FP32 work (21 flop per test) "
                               "against the FP32 peak (vector = dense FP32 MFMA = 157.3 TFLOP/s) and may exceed 1 because "
                               "the split products run at the bf16 rate; `matrix_pipe` prices what is executed")
        out["matrix_pipe"] = {"achieved": executed, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s (bf16, executed)",
                              "frac": executed / PEAK_BF16_TFLOPS, "flop_per_test": MFMA_FLOP_PER_TEST}
    if args.kernel in ("culled", "bvh"):
        # whole groups of spheres are skipped: rays x sphereCount counts tests that were never made
        out["achieved"] = out["frac"] = None
        out["note"] = (f"{args.kernel} kernel: {achieved:.1f} TFLOP/s would be the brute-force-equivalent rate (rays x "
                       "sphereCount x 21 flop), not work this kernel executes")
    return out


# ------------------------------------------------------------------ the two kinds of workload

def throughput(args, rank, world, device, backend, torch, dist, rt):
    from raytracing_amd.multi_gpu import SampleSplit

    wl = WORKLOADS[args.workload]
    w, h = wl["size"]
    scene, scene_text = make_scene(rt, wl["scene"], args.materials)
    profile = rt.PROFILE_FUSED if args.profile == "fused" else rt.PROFILE_STRICT
    kernel = {"tile": rt.KERNEL_TILE, "culled": rt.KERNEL_STREAM_CULLED, "bvh": rt.KERNEL_BVH,
              "valu": rt.KERNEL_STREAM_SGPR}.get(args.kernel, rt.KERNEL_STREAM)
    fps = args.frames_per_step

    split = SampleSplit(rank, world)
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    color = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    r = rt.Renderer(w, h, device=device)
    r.bind_buffers(accum.data_ptr(), color.data_ptr())
    r.set_profile(profile)
    r.set_kernel(kernel)
    r.set_max_depth(args.max_depth)
    r.set_camera_shortcuts(not args.no_shortcuts)
    r.set_scene(scene)

    def run_steps(n_steps):
        # every pass starts over at frame 1, so warm-up and timed steps do identical work
        for s in range(n_steps):
            split.dispatch(r, scene, step=s, frames_per_step=fps, timed=True)

    def finish(total_steps):
        """The exchange step: sum the per-rank accumulation images on rank 0 and resolve."""
        split.reduce(accum, dst=0, renderer=r)  # ordered against the renderer's stream on the device, both ways
        if rank == 0:
            r.resolve(split.divisor(total_steps, fps))

    run_steps(args.warmup)
    finish(args.warmup)
    r.timed_total()
    r.reset_counters()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    finish(args.steps)
    r.sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    kernel_ms, launches = r.timed_total()
    cnt = r.counters()
    stats = torch.tensor([elapsed, float(cnt["rays"]), float(cnt["list_rays"])], dtype=torch.float64,
                         device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    total_rays = float(stats[1])
    total_listed = float(stats[2])

    line = None
    if rank == 0:
        info = r.info()
        line = {
            "metric": wl["metric"],
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {scene_text}, {scene.sphere_count} spheres traced, {w}x{h}, "
                            f"{fps} frames (spp) per step per GPU, max depth {args.max_depth}, "
                            f"float profile {args.profile}, kernel {args.kernel}"
                            + (", material extension on" if args.materials else "")
                            + (", camera-ray shortcuts off" if args.no_shortcuts else ""),
                "parallelism": f"samples split by frame number over {world} GPU(s), one RCCL sum-reduce at the end",
                "launch": f"{info['blocks']} x {info['threads']}",
            },
            "rays_per_path": cnt["rays"] / max(1, cnt["paths"]),
            "roofline": roofline(args, cnt["sphere_tests"], launches, kernel_ms, (w // 8 * 8) * (h // 8 * 8),
                                 f"{args.workload}_{args.kernel}_{args.profile}", r.sieve_mode(),
                                 listed_share=total_listed / max(1.0, total_rays)),
        }
        if args.cpu_seconds > 0 and world == 1:  # the CPU leg runs at N=1 only
            line["cpu_baseline"] = cpu_baseline(scene, (w, h), profile, args.max_depth, args.cpu_seconds)
    r.close()
    return line


def interactive_ubos(rt, n):
    """The uniform blocks of the interactive workload's first n frames.  Every 40 frames the view
    key is tapped (acceleration along the view direction, src/application.cpp:44-47), the
    opposite key 8 frames later brakes, and at frame 16 the remaining acceleration is cancelled,
    all fed through Scene::update at dt = 1/60 s: the camera moves about one unit and comes to
    rest, so some 14 frames of each 40 restart the accumulation (frameCount = 1) and the others add
    to it.  Forward and backward taps alternate: the camera stays within a unit of where it
    started, the scene in view."""
    scene = rt.Scene.default()
    fwd = scene.camera["camera_forward"][0].copy()
    tap, brake = 60.0, 50.0
    out = []
    for i in range(n):
        k = i % 40
        way = 1.0 if (i // 40) % 2 == 0 else -1.0
        if k == 0:
            scene.add_acceleration(*(fwd * (way * tap)).tolist())
        elif k == 8:   # Scene::update scales the acceleration by 0.98 per frame
            scene.add_acceleration(*(fwd * (-way * (tap * 0.98 ** 8 + brake))).tolist())
        elif k == 16:
            scene.add_acceleration(*(fwd * (way * brake * 0.98 ** 8)).tolist())
        scene.update(1.0 / 60.0)
        out.append(scene.camera.copy())
    return scene, out


def interactive(args, rank, world, device, backend, torch, dist, rt):
    from raytracing_amd.multi_gpu import DirectTileLoop, InteractiveLoop, PeerImages, SharedFrameFlags, TileSplit

    wl = WORKLOADS["interactive"]
    w, h = wl["size"]
    profile = rt.PROFILE_FUSED if args.profile == "fused" else rt.PROFILE_STRICT
    scene, ubos = interactive_ubos(rt, args.warmup + args.steps)
    tiles = TileSplit(rank, world, h)
    colors = [torch.zeros((h, w), dtype=torch.int32, device="cuda") for _ in range(2)]
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    r = rt.Renderer(w, h, device=device)
    r.set_profile(profile)
    r.set_max_depth(args.max_depth)
    r.upload_spheres(scene.spheres)
    tiles.apply(r)
    staged = world > 1 and backend != "nccl"  # gloo rehearsal: the bands travel through host memory
    hosts = [c.cpu() for c in colors] if staged else None

    def image_of(k):
        if staged and tiles.rank != 0:
            hosts[k].copy_(colors[k])
        return hosts[k] if staged else colors[k]

    # The exchange of a frame.  "ipc": every rank's kernel writes its band straight into rank 0's colour images
    # (mapped into every process), and a counter in shared host memory says when; "nccl": the bands are sent to
    # rank 0 after the frame.  RT_BENCH_GATHER picks; "ipc" falls back to "nccl" when a rank cannot map the images.
    exchange = "none"
    direct = None
    if world > 1:
        exchange = os.environ.get("RT_BENCH_GATHER", "ipc")
        if exchange == "ipc":
            box, ok = [None], 1
            try:
                if rank == 0:
                    direct = (PeerImages(0, 0, device, w * h * 4), SharedFrameFlags(world), SharedFrameFlags(1))
                    box[0] = (direct[0].handles, direct[1].name, direct[2].name)
            except Exception as e:  # noqa: BLE001 -- any failure means "use the transfer instead"
                print(f"bench.py: rank 0 cannot export its colour images ({e})", file=sys.stderr)
            dist.broadcast_object_list(box, src=0)
            try:
                if rank != 0 and box[0] is not None:
                    direct = (PeerImages(rank, 0, device, w * h * 4, box[0][0]), SharedFrameFlags(world, box[0][1]),
                              SharedFrameFlags(1, box[0][2]))
            except Exception as e:  # noqa: BLE001
                print(f"bench.py: rank {rank} cannot map rank 0's colour images ({e})", file=sys.stderr)
            ok = torch.tensor([1 if direct is not None else 0], device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 0:  # every rank takes this branch together
                exchange = "nccl"
                if direct is not None and rank != 0:
                    direct[0].close()
                dist.barrier()
                if direct is not None:
                    if rank == 0:
                        direct[0].close()
                    direct[1].close()
                    direct[2].close()
                direct = None

    # Kernel time: two HIP events on the context's stream around the whole timed region (the launches run back
    # to back, two frames in flight, so region / launches bounds a launch from above).  Events around every
    # launch cost ~9 us of a ~45 us frame and, queued behind the previous launch, time its tail too.
    if exchange == "ipc":
        images, flags, shown = direct
        loop = DirectTileLoop(tiles, r, lambda k: r.bind_buffers_next(accum.data_ptr(), images.ptrs[k]), flags, shown)
    else:
        loop = InteractiveLoop(tiles, r, lambda k: r.bind_buffers_next(accum.data_ptr(), colors[k].data_ptr()), image_of,
                               transfer_stream=(lambda: torch.cuda.current_stream().cuda_stream) if world > 1 and not staged else None)
    for i in range(args.warmup):
        loop.frame(ubos[i])
    loop.finish()
    r.reset_counters()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.timer_begin()
    for i in range(args.steps):
        loop.frame(ubos[args.warmup + i])
    kernel_ms = r.timer_end()  # waits for the last launch
    loop.finish()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    cnt = r.counters()
    band_rows = (tiles.rows[1] - tiles.rows[0]) * 8
    stats = torch.tensor([elapsed, float(cnt["rays"]), float(cnt["sphere_tests"]), kernel_ms],
                         dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed, kernel_ms = float(tmax[0]), float(tmax[3])
    line = None
    if rank == 0:
        info = r.info()
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": wl["metric"], "value": ms, "unit": "ms", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": False, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"interactive: default scene (40 traced spheres), {w}x{h}, 1 frame per dispatch, max depth "
                            f"{args.max_depth}, float profile {args.profile}, view key tapped every 40 frames and braked, through "
                            "Scene::update (the camera moves a unit and rests: the accumulation restarts in about 14 "
                            "frames of 40), uniforms set and dispatch issued per frame",
                "parallelism": f"bands of 8x8-tile rows over {world} GPU(s); two frames in flight: frame f+1 is dispatched "
                               "before the host waits for frame f; exchange = " +
                               {"none": "none (one GPU)",
                                "ipc": "ipc: every rank's kernel writes its band into rank 0's two colour images (mapped into "
                                       "every process), a counter in shared host memory per rank says when",
                                "nccl": "nccl: the rgba8 bands of a frame are sent to rank 0 after it (two colour images)"}[exchange] +
                               "; kernel time = HIP events around the timed region / launches",
                "launch": f"{info['blocks']} x {info['threads']}",
            },
            "rays_per_second": float(stats[1]) / elapsed,
            "rays_per_path": cnt["rays"] / max(1, cnt["paths"]),
            # the slowest rank's kernel time over its launches; tests and pixels of that band
            "roofline": roofline(args, cnt["sphere_tests"], args.steps, kernel_ms, (w // 8 * 8) * band_rows,
                                 f"interactive_{args.kernel}_{args.profile}", r.sieve_mode(),
                                 listed_share=cnt["list_rays"] / max(1, cnt["rays"])),
        }
        if args.cpu_seconds > 0 and world == 1:
            line["cpu_baseline"] = cpu_baseline(scene, (w, h), profile, args.max_depth, args.cpu_seconds,
                                                interactive_frames=ubos[args.warmup:])
    r.close()
    if exchange == "ipc":  # openers unmap before the owner frees
        dist.barrier()
        if rank != 0:
            direct[0].close()
        dist.barrier()
        if rank == 0:
            direct[0].close()
        direct[1].close()
        direct[2].close()
    return line


def compact(line):
    """What `configs` keeps of a workload's line."""
    roof = line["roofline"]
    out = {"value": round(line["value"], 4 if line["unit"] == "ms" else 1), "unit": line["unit"],
           "kernel_ms": round(roof["kernel_ms_per_launch"], 4),
           "frac": None if roof["frac"] is None else round(roof["frac"], 3),
           "hbm_frac": round(roof["hbm"]["frac"], 4), "rays_per_path": round(line["rays_per_path"], 3),
           "workload": line["config"]["workload"]}
    return out


def side_configs(args, device, torch, dist, rt):
    """Short passes of the other workloads on this one GPU, after the headline's timed region: every
    configuration the baseline names gets a number from the same run of the same command."""
    out = {}
    for name, wl in WORKLOADS.items():
        if name == args.workload:
            continue
        a = copy.copy(args)
        a.workload, a.cpu_seconds, a.max_depth, a.frames_per_step = name, 0.0, wl["depth"], wl["fps"]
        a.steps, a.warmup = (200, 40) if name == "interactive" else (4, 2)
        fn = interactive if name == "interactive" else throughput
        out[name] = compact(fn(a, 0, 1, device, "nccl", torch, dist, rt))
    a = copy.copy(args)
    a.cpu_seconds, a.no_shortcuts, a.steps, a.warmup = 0.0, True, 4, 2
    brute = compact(throughput(a, 0, 1, device, "nccl", torch, dist, rt))
    brute["what"] = ("the headline workload with RT_OPTION_CAMERA_SHORTCUTS off: every Trace() call of every frame "
                     "evaluates the whole sphere list, as the shader's loop does; same bits")
    return out, brute


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s); "
                         "they must agree")

    import torch  # before the HIP library: both then share PyTorch's copy of the HIP runtime
    import torch.distributed as dist

    import raytracing_amd as rt

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU path")
    # Rehearsal hooks for a one-GPU box: RT_BENCH_DEVICE pins every rank to one device and
    # RT_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU).
    device = int(os.environ.get("RT_BENCH_DEVICE", local_rank))
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    try:
        fn = interactive if args.workload == "interactive" else throughput
        line = fn(args, rank, world, device, backend, torch, dist, rt)
        if line is not None and args.side_configs and world == 1:
            line["configs"], line["brute_force_only"] = side_configs(args, device, torch, dist, rt)
        if line is not None:
            print(json.dumps(line), flush=True)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
