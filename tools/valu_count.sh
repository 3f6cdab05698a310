#!/bin/bash
# VALU/SALU/LDS instruction counts of one headline dispatch (run on the GPU box):
#   tools/valu_count.sh [lib.so] [kernel]
lib=${1:-raytracing_amd/libraytracing_amd.so}; kernel=${2:-0}
root=$(pwd); out=$root/gpurun_out/valu_$$
cd /tmp && export TMPDIR=/tmp && cd "$root"
RT_AMD_LIB=$lib timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
    --output-format csv -d "$out" -o x -- python3 tools/one_dispatch.py $kernel 2>/dev/null | grep "^kernel"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/x_counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
    if "rt_stream" in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  ".join(f"{k} {v[-1]:.4e}" for k, v in sorted(d.items())))
PY
rm -rf "$out"
