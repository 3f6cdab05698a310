#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s at 4K on the ~500-sphere RTIOW-layout scene.

One "step" is one dispatch of --frames-per-step frames (samples per pixel) over the whole
3840x2160 image; one ray is one Trace() call of the shader (primary or bounce segment),
counted inside the kernel.

Multi-GPU (one process per GPU, launched by torch.distributed.run): samples are split by
frame number -- rank g of G renders frames g+1, g+1+G, ... with the unchanged RNG formula,
so the union of samples is the single-GPU sample set -- each rank accumulates locally and one
RCCL sum-reduce of the float4 accumulation image to rank 0 ends the job (inside the timed
region).  Scaling is weak: every rank renders `steps * frames_per_step` frames.

Output: one JSON line on rank 0.  `roofline` prices the dominant kernel against the FP32
vector peak (the kernel has no matrix contraction, so the bound is VALU, not MFMA/HBM);
`cpu_baseline` is the CPU oracle (a port of the shader, not the reference binary) on the
host cores, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 3840, 2160
SCENE_SEED = 7
FLOP_PER_TEST = 21          # algorithmic FP32 flop per ray-sphere test (a hoisted)
PEAK_FP32_TFLOPS = 157.3    # MI355X FP32 vector peak
BYTES_PER_PIXEL_DISPATCH = 36  # float4 read + float4 write + rgba8 write


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--frames-per-step", type=int, default=8)
    p.add_argument("--profile", choices=["strict", "fused"], default="strict")
    p.add_argument("--kernel", choices=["stream", "tile", "culled"], default="stream",
                   help="stream: the brute-force kernel the metric is quoted on; culled: the same kernel over a "
                        "spatially sorted sphere list, skipping groups no ray of a wave can touch (same image)")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU baseline leg (0 = skip)")
    p.add_argument("--max-depth", type=int, default=50)
    p.add_argument("--workload", choices=["rtiow4k", "interactive"], default="rtiow4k",
                   help="rtiow4k: the headline metric; interactive: 1080p, 1 frame per dispatch, moving camera, "
                        "screen tiles split across GPUs, ms/frame")
    p.add_argument("--materials", action="store_true",
                   help="rtiow4k with the material extension (lambertian/metal/dielectric per sphere) switched on; "
                        "the headline line is quoted without it (the reference has one surface response)")
    return p.parse_args()


def cpu_baseline(scene, profile, max_depth, seconds):
    """The oracle, multithreaded over tile rows, on whole 4K frames until `seconds` have passed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle as O

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores))))
    ubo = scene.camera.copy()
    acc = None
    rays = 0
    frames = 0
    t0 = time.perf_counter()
    while True:
        ubo["frameCount"][0] = frames + 1
        mats = scene.materials
        acc, _, r = O.render(ubo, scene.spheres, WIDTH, HEIGHT, max_depth=max_depth, profile=profile,
                             n_frames=1, threads=cores, accum=acc, materials=mats if mats[:, 0].any() else None)
        rays += r
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or frames >= 64:
            break
    return {
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{frames} whole frame(s) of the same {WIDTH}x{HEIGHT} scene, depth {max_depth}, "
                  f"{rays} rays in {dt:.1f} s (CPU oracle, {'fused' if profile else 'strict'} profile); "
                  + lavapipe_probe(),
    }


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


def interactive(args, rank, world, device, backend):
    """1080p, one frame per dispatch, the default scene with scripted camera impulses fed through
    Scene::update so that accumulation resets on a known schedule; every rank renders a band of
    tile rows and rank 0 gathers the rgba8 bands each frame.  A step is one displayed frame."""
    import torch
    import torch.distributed as dist

    import raytracing_amd as rt
    from raytracing_amd.multi_gpu import TileSplit

    w, h = 1920, 1080
    scene = rt.Scene.default()
    tiles = TileSplit(rank, world, h)
    color = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    r = rt.Renderer(w, h, device=device)
    r.bind_buffers(accum.data_ptr(), color.data_ptr())
    r.set_profile(rt.PROFILE_FUSED if args.profile == "fused" else rt.PROFILE_STRICT)
    r.set_max_depth(args.max_depth)
    r.upload_spheres(scene.spheres)
    tiles.apply(r)
    fwd = scene.camera["camera_forward"][0].copy()

    def frame(i):
        if i % 40 == 0:  # a key tap: one impulse along the view direction (src/application.cpp:44-47)
            scene.add_acceleration(*(fwd * 100.0).tolist())
        scene.update(1.0 / 60.0)
        r.set_uniforms(scene.camera)
        r.dispatch(1)
        r.sync()
        stage = color.cpu() if (world > 1 and backend != "nccl") else color
        tiles.gather(stage, dst=0)

    for i in range(args.warmup):
        frame(i)
    r.reset_counters()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(args.warmup + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({
            "metric": "ms/frame 1080p interactive", "value": float(t[0]) / args.steps * 1e3, "unit": "ms",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t[0]) / args.steps * 1e3,
            "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "default scene (40 spheres), 1920x1080, 1 frame per dispatch, camera impulse every "
                                   "40 frames, host sync + band gather every frame",
                       "parallelism": f"bands of 8x8-tile rows over {world} GPU(s), rgba8 bands gathered to rank 0"},
        }), flush=True)
    r.close()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist

    import raytracing_amd as rt
    from raytracing_amd.multi_gpu import SampleSplit

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

    if args.workload == "interactive":
        interactive(args, rank, world, device, backend)
        if world > 1:
            dist.destroy_process_group()
        return

    scene = rt.Scene.generate(rt.KIND_RTIOW_MATERIALS if args.materials else rt.KIND_RTIOW, 0, SCENE_SEED)
    profile = rt.PROFILE_FUSED if args.profile == "fused" else rt.PROFILE_STRICT
    kernel = {"tile": rt.KERNEL_TILE, "culled": rt.KERNEL_STREAM_CULLED}.get(args.kernel, rt.KERNEL_STREAM)
    fps = args.frames_per_step

    split = SampleSplit(rank, world)
    accum = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device="cuda")
    color = torch.zeros((HEIGHT, WIDTH), dtype=torch.int32, device="cuda")
    r = rt.Renderer(WIDTH, HEIGHT, device=device)
    r.bind_buffers(accum.data_ptr(), color.data_ptr())
    r.set_profile(profile)
    r.set_kernel(kernel)
    r.set_max_depth(args.max_depth)
    r.set_scene(scene)

    def run_steps(n_steps):
        # every pass starts over at frame 1, so warm-up and timed steps do identical work
        for s in range(n_steps):
            split.dispatch(r, scene, step=s, frames_per_step=fps)

    def finish(total_steps):
        """The exchange step: sum the per-rank accumulation images on rank 0 and resolve."""
        r.sync()
        split.reduce(accum, dst=0)
        if rank == 0:
            torch.cuda.current_stream().synchronize()
            r.resolve(split.divisor(total_steps, fps))

    run_steps(args.warmup)
    finish(args.warmup)
    r.sync()
    r.reset_counters()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.timer_begin()
    run_steps(args.steps)
    kernel_ms = r.timer_end()
    finish(args.steps)
    r.sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    cnt = r.counters()
    stats = torch.tensor([elapsed, float(cnt["rays"]), float(cnt["sphere_tests"])], dtype=torch.float64,
                         device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    total_rays = float(stats[1])

    if rank == 0:
        launch_ms = kernel_ms / args.steps
        tests_per_launch = cnt["sphere_tests"] / args.steps
        achieved = tests_per_launch * FLOP_PER_TEST / (launch_ms * 1e-3) / 1e12
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(f"{args.kernel}_{args.profile}_fps{fps}")
            except Exception:
                traffic = None
        info = r.info()
        line = {
            "metric": "Mrays/sec (whole node) @4K ~500 spheres",
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
                "workload": f"rtiow-layout scene seed {SCENE_SEED} ({scene.sphere_count} spheres), {WIDTH}x{HEIGHT}, "
                            f"{fps} frames (spp) per step per GPU, max depth {args.max_depth}, "
                            f"float profile {args.profile}, kernel {args.kernel}"
                            + (", material extension on" if args.materials else ""),
                "parallelism": f"samples split by frame number over {world} GPU(s), one RCCL sum-reduce at the end",
                "launch": f"{info['blocks']} x {info['threads']}",
            },
            "rays_per_path": cnt["rays"] / max(1, cnt["paths"]),
            "roofline": {
                "bound": "valu",
                "bound_detail": "FP32 vector ALU: the path has no matrix contraction, so neither HBM nor MFMA bounds it; "
                                "the FP32 vector peak equals the dense FP32 MFMA peak (157.3 TFLOP/s)",
                "achieved": achieved,
                "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS,
                "traffic": traffic,
                "kernel_ms_per_launch": launch_ms,
                "flop_per_test": FLOP_PER_TEST,
                "hbm_algorithmic_GBps": WIDTH // 8 * 8 * (HEIGHT // 8 * 8) * BYTES_PER_PIXEL_DISPATCH
                                        / (launch_ms * 1e-3) / 1e9,
            },
        }
        if args.cpu_seconds > 0 and world == 1:  # the CPU leg runs at N=1 only
            line["cpu_baseline"] = cpu_baseline(scene, profile, args.max_depth, args.cpu_seconds)
        print(json.dumps(line), flush=True)

    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
