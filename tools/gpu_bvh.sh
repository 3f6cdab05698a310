#!/bin/bash
# GPU-box call for the per-lane traversal: its parity tests, step counts, then timings against the other kernels.
set -e
root=$(pwd); out=$root/gpurun_out/${1:-bvh}; mkdir -p "$out"
timeout -k 10 600 python -u -m pytest tests/test_gpu_culled.py -x -q --timeout 300 --timeout-method thread > "$out/tests.log" 2>&1 || { tail -40 "$out/tests.log"; exit 1; }
tail -3 "$out/tests.log"
if [ -f variants/libpasses.so ]; then
  for wl in "rtiow 3840 2160 8 50 8 1" "cube4096 3840 2160 2 50 8 1" "cube1000 3840 2160 4 50 8 1"; do
    RT_AMD_LIB=variants/libpasses.so timeout -k 10 120 python3 tools/one_dispatch.py $wl 2>&1 | grep -v "^passes" | tee -a "$out/s.txt"
  done
fi
for k in 0 8; do
  for wl in "cube16384 3840 2160 2 50 $k 2" "cube65536 3840 2160 1 50 $k 1"; do
    timeout -k 10 200 python3 tools/one_dispatch.py $wl 2>&1 | tee -a "$out/k.txt"
  done
done
for k in 0 4 8; do
  for wl in "rtiow 3840 2160 8 50 $k 3" "cube4096 3840 2160 2 50 $k 3" "cube1000 3840 2160 4 50 $k 3"; do
    timeout -k 10 120 python3 tools/one_dispatch.py $wl 2>&1 | tee -a "$out/k.txt"
  done
done
