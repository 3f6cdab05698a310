#!/bin/bash
# One GPU-box call: the whole GPU test suite, then every bench workload.  usage: tools/gpu_check.sh TAG
set -e
root=$(pwd); out=$root/gpurun_out/${1:-check}; mkdir -p "$out"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$out/gpu_tests.log" 2>&1 || { tail -40 "$out/gpu_tests.log"; exit 1; }
tail -3 "$out/gpu_tests.log"
for wl in rtiow4k default720p_d8 default1080p stress4096 interactive; do
  timeout -k 10 300 python3 bench.py --workload $wl > "$out/bench_$wl.json" 2> "$out/bench_$wl.err" || { tail -20 "$out/bench_$wl.err"; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$out/bench_$wl.json')); print('$wl', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roofline', d.get('roofline'))"
done
