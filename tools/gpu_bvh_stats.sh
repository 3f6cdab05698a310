#!/bin/bash
# Per-lane traversal: step counts and phase cycles (experiment builds under variants/).
set -e
root=$(pwd); out=$root/gpurun_out/${1:-bvhstats}; mkdir -p "$out"
for lib in passes timers; do
  for wl in "rtiow 3840 2160 8 50 8 1" "cube4096 3840 2160 2 50 8 1" "cube1000 3840 2160 4 50 8 1" "rtiow 3840 2160 8 50 4 1"; do
    echo "== $lib: $wl" | tee -a "$out/s.txt"
    RT_AMD_LIB=variants/lib$lib.so timeout -k 10 120 python3 tools/one_dispatch.py $wl 2>&1 | tee -a "$out/s.txt"
  done
done
