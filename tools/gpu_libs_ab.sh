#!/bin/bash
# Development aid: builds against each other, alternating.  usage: tools/gpu_libs_ab.sh TAG "ENV=.. ENV=.." lib1.so lib2.so ...
set -e
root=$(pwd); tag=$1; envs=$2; shift 2; out=$root/gpurun_out/$tag; mkdir -p "$out"
IFS=';' read -ra WLS <<< "${RT_AB_WL:-rtiow 3840 2160 8 50 0 40}"
for rep in 1 2 3; do
  for lib in "$@"; do
    for wl in "${WLS[@]}"; do
      echo -n "$(basename $lib .so): " | tee -a "$out/k.txt"
      env $envs RT_AMD_LIB=$lib timeout -k 10 120 python3 tools/one_dispatch.py $wl 2>&1 | tee -a "$out/k.txt"
    done
  done
done
