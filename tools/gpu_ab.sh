#!/bin/bash
# A/B of library builds on a few dispatch shapes.  usage: tools/gpu_ab.sh TAG lib1.so lib2.so ... (RT_AB_WL overrides the workloads, ';'-separated)
set -e
root=$(pwd); tag=$1; shift; out=$root/gpurun_out/$tag; mkdir -p "$out"
IFS=';' read -ra WLS <<< "${RT_AB_WL:-rtiow 3840 2160 8 50 0 5;cube4096 3840 2160 2 50 0 3;cube1000 3840 2160 4 50 0 3}"
for rep in 1 2; do
for lib in "$@"; do
  for wl in "${WLS[@]}"; do
    echo -n "$lib: " | tee -a "$out/k.txt"
    RT_AMD_LIB=$lib timeout -k 10 120 python3 tools/one_dispatch.py $wl 2>&1 | tee -a "$out/k.txt"
  done
done
done
