#!/usr/bin/env python3
"""Turn the output of tools/collect_profiles.sh into the committed summaries under profiles/.

usage: python tools/summarise_profiles.py TAG WORKLOAD [WORKLOAD ...]

Per workload: profiles/<TAG>_<workload>_kernel_stats.csv (rocprofv3 --kernel-trace --stats),
profiles/<TAG>_<workload>_bench_line.json (the JSON line of that profiled run and of an
unprofiled run of the same command) and profiles/<TAG>_<workload>_counters.json (per-launch means
of every counter collected).  All workloads: profiles/<TAG>_pmc.json, keyed
<workload>_<kernel>_<profile>, which bench.py reads for roofline.traffic, valu_busy, clock_ghz and
executed_flop_frac.

Units and corrections: FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts half the
bytes of wide coalesced reads, so it is doubled.  GRBM_GUI_ACTIVE is summed over the 8 XCDs:
clock = GRBM_GUI_ACTIVE / 8 / kernel time.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over
the SIMDs: VALU busy = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 1024
PEAK_TFLOPS = 157.3
EXECUTED_FLOP_PER_TEST = 15.0  # the sieve as compiled: 7 v_pk_fma_f32 + 1 v_pk_add_f32 per sphere PAIR = 30 flop / 2
KERNELS = ("rt_stream_kernel", "rt_tile_kernel", "rt_lite_kernel")


def per_launch(path, timed):
    """{counter: mean over the last `timed` launches}, mean duration (ns) of those launches, kernel name.
    A step of the stream kernel also launches small kernels of the library (rt_sky_frames_kernel for the pixels that
    see the sky, the join of split tiles, the tile order every eighth step): the bytes they move belong to the step,
    so FETCH_SIZE and WRITE_SIZE include theirs (summed over the window of the timed launches, per launch);
    every other counter is the main kernel's alone."""
    every = list(csv.DictReader(open(path)))
    rows = [r for r in every if any(k in r["Kernel_Name"] for k in KERNELS)]
    by = collections.defaultdict(list)
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        by[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                      int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out, dur = {}, None
    for name, v in by.items():
        window_from = v[-timed - 1][3] if len(v) > timed else 0
        window_to = v[-1][3]
        v = v[-timed:]
        out[name] = sum(x[0] for x in v) / len(v)
        dur = sum(x[1] for x in v) / len(v)
        if name in ("FETCH_SIZE", "WRITE_SIZE"):
            helpers = [r for r in every if r["Counter_Name"] == name and "rt_" in r["Kernel_Name"]
                       and not any(k in r["Kernel_Name"] for k in KERNELS)
                       and window_from < int(r["Start_Timestamp"]) <= window_to]
            out[name] += sum(float(r["Counter_Value"]) for r in helpers) / len(v)
    return out, dur, (rows[0]["Kernel_Name"].split("(")[0] if rows else None)


def smi_under_load(path):
    """Shader clock (MHz) and socket power (W) that rocm-smi reported while a long run of the workload was
    going: the samples taken once the clock had left its idle value, median and extremes."""
    import re
    import statistics

    if not os.path.exists(path):
        return None
    clocks, power = [], []
    for ln in open(path):
        ln = ln.strip()
        if not ln.startswith("{"):
            continue
        try:
            card = next(iter(json.loads(ln).values()))
        except Exception:
            continue
        for k, v in card.items():
            m = re.search(r"\((\d+)Mhz\)", str(v))
            if "sclk" in k.lower() and m:
                clocks.append(int(m.group(1)))
            if "power" in k.lower() and "(w)" in k.lower():
                try:
                    power.append(float(v))
                except ValueError:
                    pass
    busy = [(c, p) for c, p in zip(clocks, power)] if len(clocks) == len(power) else [(c, None) for c in clocks]
    if power:
        busy = [(c, p) for c, p in busy if p is not None and p > 0.5 * max(power)] or busy
    elif clocks:  # no power reading for this user: the samples whose clock has left the idle value
        busy = [(c, None) for c in clocks if c > 0.5 * max(clocks)]
    if not busy:
        return None
    cs = [c for c, _ in busy]
    ps = [p for _, p in busy if p is not None]
    return {"samples": len(busy), "sclk_mhz_median": statistics.median(cs), "sclk_mhz_min": min(cs), "sclk_mhz_max": max(cs),
            "power_w_median": statistics.median(ps) if ps else None, "power_w_max": max(ps) if ps else None,
            "how": "rocm-smi -c -P every ~0.7 s during a run of the same command with more steps; samples above half the peak power (or, without a power reading, half the peak clock)"}


def one(tag, wl, pmc_all):
    src = os.path.join(ROOT, "gpurun_out", "prof", tag, wl)
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv"))
    line = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1])
    plain = None
    try:
        plain = json.loads(open(os.path.join(src, "bench_line_plain.json")).read().strip().splitlines()[-1])
    except Exception:
        pass
    json.dump({"command": "python3 bench.py " + open(os.path.join(src, "bench_args.txt")).read().strip(),
               "under_rocprofv3_kernel_trace": line, "unprofiled": plain},
              open(os.path.join(dst, f"{tag}_{wl}_bench_line.json"), "w"), indent=1)
    if os.environ.get("RT_SUMMARISE_LINES_ONLY"):  # second pass: counters and <tag>_pmc.json stay as the first pass left them
        return
    timed = int(line["roofline"]["launches"])
    counters, durations, kernel = {}, {}, None
    for p in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_wait"):
        files = glob.glob(os.path.join(src, p, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        c, dur, k = per_launch(files[0], timed)
        kernel = kernel or k
        counters.update(c)
        for name in c:
            durations[name] = dur
    rec = {"_comment": f"per-launch means over the {timed} timed launches of `{kernel}` in `python3 bench.py "
                       + open(os.path.join(src, "bench_args.txt")).read().strip() + "`, one rocprofv3 --pmc pass per "
                       "counter group (tools/collect_profiles.sh); *_ns = mean launch duration in the pass that counter came from",
           "counters": counters, "duration_ns": durations}
    json.dump(rec, open(os.path.join(dst, f"{tag}_{wl}_counters.json"), "w"), indent=1)

    # a label "<workload>_<kernel>" comes from a run with --kernel <kernel>; bench.py's key is <workload>_<kernel>_<profile>
    known = [k for k in ("bvh", "culled", "valu", "tile") if wl.endswith("_" + k)]
    key = f"{wl}_strict" if known else f"{wl}_stream_strict"
    out = {}
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        out["fetch_bytes_per_launch"] = counters["FETCH_SIZE"] * 1024 * 2
        out["write_bytes_per_launch"] = counters["WRITE_SIZE"] * 1024
        out["traffic_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
    if "GRBM_GUI_ACTIVE" in counters:
        cyc = counters["GRBM_GUI_ACTIVE"] / 8.0
        out["clock_ghz"] = cyc / durations["GRBM_GUI_ACTIVE"]
        if "SQ_ACTIVE_INST_VALU" in counters:
            out["valu_busy"] = 4.0 * counters["SQ_ACTIVE_INST_VALU"] / N_SIMD / cyc
        if "SQ_INSTS_VALU" in counters:
            out["valu_insts_per_launch"] = counters["SQ_INSTS_VALU"]
        if counters.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            # cycles, summed over the SIMDs, in which the matrix pipe works (32 per v_mfma_f32_32x32x16_bf16)
            out["mfma_busy"] = counters["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD / (durations["SQ_VALU_MFMA_BUSY_CYCLES"] * out["clock_ghz"])
            out["mfma_insts_per_launch"] = counters.get("SQ_INSTS_MFMA")
    if line["roofline"]["bound"] == "valu" and line["roofline"]["achieved"] is not None:
        tests_per_launch = (line["roofline"]["achieved"] * 1e12 * line["roofline"]["kernel_ms_per_launch"] * 1e-3
                            / line["roofline"]["flop_per_test"])
        out["executed_flop_frac"] = (tests_per_launch * EXECUTED_FLOP_PER_TEST / (line["roofline"]["kernel_ms_per_launch"] * 1e-3)
                                     / 1e12 / PEAK_TFLOPS)
        out["executed_flop_note"] = ("sieve on the vector ALU only: 15 FP32 flop executed per ray-sphere test "
                                     "(7 v_pk_fma_f32 + 1 v_pk_add_f32 per sphere pair)")
    out["kernel_ms_per_launch_rocprof_stats"] = None
    for r in csv.DictReader(open(os.path.join(src, "stats", "bench_kernel_stats.csv"))):
        if any(k in r["Name"] for k in KERNELS):
            out["kernel_ms_per_launch_rocprof_stats"] = float(r["AverageNs"]) / 1e6
            out["rocprof_calls"] = int(r["Calls"])
            break
    out["kernel_ms_per_launch_bench"] = line["roofline"]["kernel_ms_per_launch"]
    smi = smi_under_load(os.path.join(src, "smi_samples.jsonl"))
    if smi:
        out["smi_under_load"] = smi
        if "clock_ghz" in out:
            out["fp32_peak_at_counter_clock_tflops"] = PEAK_TFLOPS * out["clock_ghz"] / 2.4
    pmc_all[key] = out
    print(wl, json.dumps(out, indent=1))


def main():
    tag = sys.argv[1]
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc.json")
    pmc_all = json.load(open(path)) if os.path.exists(path) else {}
    pmc_all["_comment"] = ("per workload (key <workload>_<kernel>_<profile>): HBM bytes per launch from FETCH_SIZE (doubled: gfx950 "
                           "reports half the bytes of wide coalesced reads) + WRITE_SIZE; clock = GRBM_GUI_ACTIVE / 8 XCDs / launch "
                           "time; valu_busy = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8); "
                           "tools/collect_profiles.sh + tools/summarise_profiles.py")
    for wl in sys.argv[2:]:
        one(tag, wl, pmc_all)
    json.dump(pmc_all, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
