#!/bin/bash
# Development aid: the stream kernel's frame split of heavy tiles against no split and against another build.
#   tools/gpu_split_ab.sh TAG   (variants/libhead.so = the build to compare with, variants/libknobs.so = this tree)
set -e
root=$(pwd); tag=$1; out=$root/gpurun_out/$tag; mkdir -p "$out"
IFS=';' read -ra WLS <<< "${RT_AB_WL:-rtiow 3840 2160 8 50 0 8;cube4096 3840 2160 2 50 0 4;rtiow 1920 1080 8 50 0 16;cube1000 3840 2160 4 50 0 4}"
run() {  # label, lib, env...
  label=$1; lib=$2; shift 2
  for wl in "${WLS[@]}"; do
    echo -n "$label: " | tee -a "$out/k.txt"
    env "$@" RT_AMD_LIB=$lib timeout -k 10 120 python3 tools/one_dispatch.py $wl 2>&1 | tee -a "$out/k.txt"
  done
}
for rep in 1 2; do
  run head variants/libhead.so X=1
  run nosplit variants/libknobs.so RT_DEBUG_STREAM_SPLIT_DEN=0
  for cfg in ${RT_AB_CFG:-4:8 2:8 8:8 4:4 4:2}; do
    IFS=: read -r den ways tiles <<< "$cfg"
    run "den${den}_ways${ways}_tiles${tiles:-24}" variants/libknobs.so RT_DEBUG_STREAM_SPLIT_DEN=$den RT_DEBUG_STREAM_SPLIT_WAYS=$ways RT_DEBUG_STREAM_SPLIT_TILES=${tiles:-24}
  done
done
