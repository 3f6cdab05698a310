#!/bin/bash
set -e
root=$(pwd); out=$root/gpurun_out/r2d; mkdir -p "$out"
export RT_DEBUG_STATIC_EIGHTHS=0
for pad in 0 45000 70000; do
  echo "== LDS pad $pad (timers build)" | tee -a "$out/pad.txt"
  RT_DEBUG_LDS_PAD=$pad RT_AMD_LIB=variants/libtimers.so timeout -k 10 120 python3 tools/one_dispatch.py default 1920 1080 1 50 0 100 40 2 2>&1 | tee -a "$out/pad.txt"
  echo "== LDS pad $pad (product build)" | tee -a "$out/pad.txt"
  RT_DEBUG_LDS_PAD=$pad timeout -k 10 120 python3 tools/one_dispatch.py default 1920 1080 1 50 0 200 40 2 2>&1 | tee -a "$out/pad.txt"
  RT_DEBUG_LDS_PAD=$pad timeout -k 10 120 python3 tools/one_dispatch.py default 1280 720 64 8 0 5 40 1 2>&1 | tee -a "$out/pad.txt"
done
