#!/bin/bash
set -e
root=$(pwd); out=$root/gpurun_out/${1:-far2}; mkdir -p "$out"
for lib in raytracing_amd/libraytracing_amd.so variants/libahead4.so variants/libahead8.so; do
  for wl in "cube4096 3840 2160 2 50 0 3" "cube2000 3840 2160 4 50 0 3"; do
    echo "== $lib" | tee -a "$out/k.txt"
    RT_AMD_LIB=$lib timeout -k 10 120 python3 tools/one_dispatch.py $wl 2>&1 | tee -a "$out/k.txt"
  done
done
