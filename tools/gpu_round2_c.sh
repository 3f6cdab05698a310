#!/bin/bash
set -e
root=$(pwd); out=$root/gpurun_out/r2c; mkdir -p "$out"
export RT_DEBUG_STATIC_EIGHTHS=0
for wl in "1920 1080 1 50 0 200 0 2" "1920 1080 1 50 0 200 8 2" "1920 1080 1 50 0 200 40 2" "1920 1080 1 50 0 200 40 1" "1920 1080 1 1 0 200 40 2" \
          "3840 2160 1 50 0 50 0 2" "3840 2160 1 50 0 50 40 2" "1920 1080 4 50 0 50 40 2" "1920 1080 4 50 0 50 0 2" "1920 1080 1 50 1 200 40 2" "1920 1080 1 50 1 200 0 2"; do
  timeout -k 10 120 python3 tools/one_dispatch.py default $wl | tee -a "$out/decomp.txt"
done
