#!/bin/bash
# One GPU-box call: tools/collect_profiles.sh for every bench workload, and for the two kernels that are not
# brute force on the two large scenes.  usage: tools/gpu_profiles.sh TAG [workload ...]
set -e
tag=${1:-r02}; shift || true
wls=${*:-rtiow4k default720p_d8 default1080p stress4096 interactive}
for wl in $wls; do
  bash tools/collect_profiles.sh "$tag" "$wl"
done
if [ -z "$RT_PROFILES_BRUTE_ONLY" ]; then
  for wl in rtiow4k stress4096; do
    for k in culled bvh; do
      bash tools/collect_profiles.sh "$tag" "$wl" --kernel $k
    done
  done
fi
