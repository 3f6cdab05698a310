#!/usr/bin/env python3
"""Turn the output of tools/collect_profiles.sh into the committed summaries under profiles/.

usage: python tools/summarise_profiles.py TAG [steps=8]

Writes profiles/<TAG>_bench_kernel_stats.csv, profiles/<TAG>_bench_line.json and updates
profiles/pmc_traffic.json (HBM bytes per launch of the dominant kernel, which bench.py reports
as roofline.traffic).  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half
the bytes of wide coalesced reads, so it is doubled (see the comment in the JSON).
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_per_launch(path, kernel, counter, steps):
    rows = [r for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    vals = [float(r["Counter_Value"]) for r in rows][-steps:]  # the timed launches are the last ones
    return sum(vals) / len(vals), vals


def main():
    tag = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    src = os.path.join(ROOT, "gpurun_out", "prof", tag)
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    line = json.load(open(os.path.join(src, "bench_line.json")))
    json.dump(line, open(os.path.join(dst, f"{tag}_bench_line.json"), "w"), indent=1)
    kernel = "rt_stream_kernel"
    fetch_kib, fv = counter_per_launch(os.path.join(src, "pmc_fetch", "bench_counter_collection.csv"), kernel, "FETCH_SIZE", steps)
    write_kib, wv = counter_per_launch(os.path.join(src, "pmc_write", "bench_counter_collection.csv"), kernel, "WRITE_SIZE", steps)
    fetch = fetch_kib * 1024 * 2
    write = write_kib * 1024
    px = 3840 * 2160
    out = {
        "_comment": "HBM bytes per launch of rt_stream_kernel in `python bench.py` (8 frames per step, 3840x2160): "
                    "mean over the timed launches of FETCH_SIZE (KiB, doubled: gfx950 reports half the bytes of a "
                    "wide coalesced read) + WRITE_SIZE (KiB), collected in separate rocprofv3 --pmc passes "
                    f"(tools/collect_profiles.sh, tag {tag}). Algorithmic: 16 B read (except in the clearing first "
                    "step of a pass) + 16 B accumulation + 4 B rgba8 written per pixel "
                    f"= {px * 36 / 1e6:.0f} MB ({px * 20 / 1e6:.0f} MB for the clearing step).",
        "stream_strict_fps8": fetch + write,
        "fetch_bytes": fetch,
        "write_bytes": write,
        "per_launch_fetch_KiB": fv,
        "per_launch_write_KiB": wv,
    }
    json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
    for r in csv.DictReader(open(os.path.join(src, "stats", "bench_kernel_stats.csv"))):
        if kernel in r["Name"]:
            print("kernel stats:", r["Calls"], "calls, average", float(r["AverageNs"]) / 1e6, "ms; bench.py measured",
                  line["roofline"]["kernel_ms_per_launch"], "ms")


if __name__ == "__main__":
    main()
