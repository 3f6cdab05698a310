#!/bin/bash
set -e
root=$(pwd); out=$root/gpurun_out/r2g; mkdir -p "$out"
for k in 2 5; do
 for wl in "default 1920 1080 1 50 $k 200 40 2" "default 1280 720 64 8 $k 5" "default 1920 1080 64 50 $k 3" "rtiow 3840 2160 8 50 $k 3" "cube1000 3840 2160 4 50 $k 3" "cube200 3840 2160 8 50 $k 3"; do
  timeout -k 10 120 python3 tools/one_dispatch.py $wl | tee -a "$out/k.txt"
 done
done
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$out/gpu_tests.log" 2>&1 || { tail -40 "$out/gpu_tests.log"; exit 1; }
tail -3 "$out/gpu_tests.log"
