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


def test_bench_single_rank_line():
    res = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--frames-per-step", "2",
                          "--cpu-seconds", "1"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    line = _one_json_line(res.stdout)
    assert KEYS <= set(line) and "cpu_baseline" in line
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["unit"] == "Mrays/s" and line["value"] > 1000.0 and line["higher_is_better"] is True
    roof = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof)
    assert 0.0 < roof["frac"] < 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = line["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu) and cpu["kind"] == "port" and cpu["value"] > 0
    assert "workload" in line["config"] and "model" not in line["config"]


@pytest.mark.parametrize("workload", ["rtiow4k", "interactive"])
def test_bench_two_ranks_rehearsal(workload):
    env = dict(os.environ, RT_BENCH_DEVICE="0", RT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = "29617" if workload == "rtiow4k" else "29618"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", port, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--frames-per-step", "2", "--cpu-seconds", "0", "--workload", workload]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = _one_json_line(res.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "cpu_baseline" not in line  # the CPU leg runs at one rank only
    if workload == "rtiow4k":
        assert line["scaling"] == "weak" and line["unit"] == "Mrays/s"
    else:
        assert line["unit"] == "ms" and line["higher_is_better"] is False
