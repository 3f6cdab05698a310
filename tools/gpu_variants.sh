#!/bin/bash
# One GPU-box call: the headline workload under its options, one line each.  usage: tools/gpu_variants.sh TAG
set -e
root=$(pwd); out=$root/gpurun_out/${1:-variants}; mkdir -p "$out"; : > "$out/bench_variants.txt"
run() {
  echo -n "bench.py $*: " >> "$out/bench_variants.txt"
  timeout -k 10 300 python3 bench.py "$@" --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value'],1), d['unit'], 'ms/step', round(d['ms_per_step'],3), 'kernel ms', round(r['kernel_ms_per_launch'],3), 'frac', r['frac'] and round(r['frac'],3))" >> "$out/bench_variants.txt"
}
run --workload rtiow4k
run --workload rtiow4k --no-shortcuts
run --workload rtiow4k --profile fused
run --workload rtiow4k --materials
run --workload rtiow4k --kernel bvh --materials
run --workload rtiow4k --kernel valu
run --workload rtiow4k --kernel culled
run --workload rtiow4k --kernel bvh
run --workload stress4096
run --workload stress4096 --no-shortcuts
run --workload stress4096 --kernel valu
run --workload stress4096 --kernel bvh
run --workload rtiow4k --steps 600
cat "$out/bench_variants.txt"
