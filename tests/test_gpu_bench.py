"""GPU test of the benchmark contract: bench.py prints exactly one JSON line with the agreed keys,
at one rank and -- rehearsed on one GPU with gloo standing in for RCCL -- at two ranks launched the
way the driver launches them."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _check_roofline(line, culled=False):
    roof = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "hbm"} <= set(roof)
    if not culled:
        # the algorithmic FP32 fraction may pass 1 once the contraction runs on the matrix cores
        assert 0.0 < roof["frac"] < (1.0 if roof["bound"] == "valu" else 3.0)
        assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    if roof["bound"] == "mfma":
        mp = roof["matrix_pipe"]
        assert 0.0 < mp["frac"] < 1.0 and abs(mp["frac"] - mp["achieved"] / mp["peak"]) < 1e-9
    hbm = roof["hbm"]
    assert hbm["unit"] == "GB/s" and 0.0 < hbm["frac"] < 1.0 and abs(hbm["frac"] - hbm["achieved"] / hbm["peak"]) < 1e-9
    assert roof["launches"] == line["steps"] and roof["kernel_ms_per_launch"] > 0
    # device time of the kernel launches cannot exceed the wall time of the steps
    assert roof["kernel_ms_per_launch"] <= line["ms_per_step"] * 1.001
    assert (roof["traffic"] is None) == (roof["traffic_source"] is None)


@pytest.mark.parametrize("workload,unit", [("rtiow4k", "Mrays/s"), ("default720p_d8", "Mrays/s"),
                                           ("default1080p", "Mrays/s"), ("stress4096", "Mrays/s"),
                                           ("interactive", "ms")])
def test_bench_single_rank_line(workload, unit):
    steps = "20" if workload == "interactive" else "2"
    cmd = [sys.executable, "bench.py", "--workload", workload, "--steps", steps, "--warmup", "1", "--cpu-seconds", "1"]
    if workload != "interactive":
        cmd += ["--frames-per-step", "2"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    line = _one_json_line(res.stdout)
    assert KEYS <= set(line) and "cpu_baseline" in line
    assert line["n_gpus"] == 1 and line["steps"] == int(steps) and line["warmup"] == 1
    assert line["unit"] == unit and line["value"] > 0
    if unit == "Mrays/s":
        assert line["higher_is_better"] is True and line["scaling"] == "weak"
        assert line["value"] > (1000.0 if workload != "stress4096" else 100.0)
    else:
        assert line["higher_is_better"] is False and line["value"] < 50.0
        assert line["rays_per_path"] > 1.2  # the default scene in view: a camera looking at sky traces 1.0
    _check_roofline(line)
    cpu = line["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu) and cpu["kind"] == "port" and cpu["value"] > 0
    assert cpu["unit"] == ("ms/frame" if unit == "ms" else "Mrays/s")
    assert workload in line["config"]["workload"] and "model" not in line["config"]


def test_bench_default_line_carries_every_configuration():
    """The command the driver runs (no --workload): the headline workload, and measured after its timed
    region a short pass of each other configuration and of the headline with the camera-ray shortcuts off."""
    res = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    line = _one_json_line(res.stdout)
    assert KEYS <= set(line) and "rtiow4k" in line["config"]["workload"] and line["steps"] == 2
    _check_roofline(line)
    assert set(line["configs"]) == {"default720p_d8", "default1080p", "stress4096", "interactive"}
    for name, c in line["configs"].items():
        assert {"value", "unit", "kernel_ms", "frac", "hbm_frac"} <= set(c) and c["value"] > 0 and name in c["workload"], name
        assert c["unit"] == ("ms" if name == "interactive" else "Mrays/s")
    assert line["configs"]["interactive"]["value"] < 1.0  # ms per 1080p frame
    brute = line["brute_force_only"]
    assert brute["unit"] == "Mrays/s" and 0 < brute["value"] < line["value"] and "shortcuts off" in brute["workload"]
    # with every test evaluated the algorithmic rate is rays x spheres x 21 flop over the kernel time
    assert brute["frac"] > line["roofline"]["frac"]


@pytest.mark.parametrize("kernel", ["culled", "bvh"])
def test_bench_culled_kernel_does_not_claim_brute_force_flops(kernel):
    res = subprocess.run([sys.executable, "bench.py", "--kernel", kernel, "--steps", "2", "--warmup", "1",
                          "--frames-per-step", "2", "--cpu-seconds", "0"], cwd=ROOT, capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    line = _one_json_line(res.stdout)
    assert line["roofline"]["frac"] is None and "brute-force-equivalent" in line["roofline"]["note"]
    _check_roofline(line, culled=True)


@pytest.mark.parametrize("launcher", ["torchrun", "plain"])
@pytest.mark.parametrize("workload", ["rtiow4k", "interactive"])
def test_bench_two_ranks_rehearsal(workload, launcher):
    """Two ranks on one GPU with gloo standing in for RCCL, started the way the driver starts them
    (torch.distributed.run) and by plain `python bench.py --gpus 2`, which spawns its ranks itself."""
    env = dict(os.environ, RT_BENCH_DEVICE="0", RT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    port = "29617" if workload == "rtiow4k" else "29618"
    tail = ["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--cpu-seconds", "0", "--workload", workload]
    if workload != "interactive":
        tail += ["--frames-per-step", "2"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", port] + tail
    else:
        cmd = [sys.executable] + tail
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = _one_json_line(res.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "cpu_baseline" not in line  # the CPU leg runs at one rank only
    if workload == "rtiow4k":
        assert line["scaling"] == "weak" and line["unit"] == "Mrays/s"
    else:
        assert line["unit"] == "ms" and line["higher_is_better"] is False and line["scaling"] == "strong"
        assert "exchange = ipc" in line["config"]["parallelism"]  # the two ranks map rank 0's images (one device here)


def test_bench_interactive_two_ranks_with_the_nccl_style_gather():
    """The fallback exchange of the interactive workload (bands sent to rank 0 after every frame), rehearsed with gloo."""
    env = dict(os.environ, RT_BENCH_DEVICE="0", RT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", RT_BENCH_GATHER="nccl")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--cpu-seconds", "0",
                          "--workload", "interactive"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = _one_json_line(res.stdout)
    assert line["n_gpus"] == 2 and "exchange = nccl" in line["config"]["parallelism"]


def test_bench_refuses_a_world_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "must agree" in res.stderr
