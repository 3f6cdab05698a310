#!/usr/bin/env python3
"""Per-dispatch means of every counter rocprofv3 collected under DIR/pmc*_<tag>/ (development aid).
usage: tools/pmc_table.py DIR [kernel-name-substring]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "rt_"
for d in sorted(glob.glob(os.path.join(root, "pmc*_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print(os.path.basename(d), "no counter file")
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        if want in r["Kernel_Name"]:
            per[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        # the first dispatch of a run is the warm-up
        print(os.path.basename(d), k, " ".join(f"{n}={sum(v[1:]) / max(1, len(v) - 1):.4g}" for n, v in sorted(cs.items())),
              f"n={len(next(iter(cs.values())))}")
