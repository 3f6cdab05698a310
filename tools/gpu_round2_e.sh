#!/bin/bash
set -e
root=$(pwd); out=$root/gpurun_out/r2e; mkdir -p "$out"
timeout -k 10 900 python -u -m pytest tests/test_gpu_bench.py -x -q --timeout 300 --timeout-method thread > "$out/bench_tests.log" 2>&1 || { tail -40 "$out/bench_tests.log"; exit 1; }
tail -3 "$out/bench_tests.log"
for wl in rtiow4k default720p_d8 default1080p stress4096 interactive; do
  bash tools/collect_profiles.sh r02 $wl
done
