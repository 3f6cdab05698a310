#!/bin/bash
# Matrix-core sieve with fragments from global memory (scenes of 1025..4096 spheres): parity tests, then timings.
set -e
root=$(pwd); out=$root/gpurun_out/${1:-far}; mkdir -p "$out"
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > "$out/tests.log" 2>&1 || { tail -40 "$out/tests.log"; exit 1; }
tail -3 "$out/tests.log"
for wl in "cube4096 3840 2160 2 50 0 3" "cube2000 3840 2160 4 50 0 3" "cube1000 3840 2160 4 50 0 3"; do
  RT_DEBUG_MFMA_FAR=0 timeout -k 10 120 python3 tools/one_dispatch.py $wl 2>&1 | tee -a "$out/k.txt"
  timeout -k 10 120 python3 tools/one_dispatch.py $wl 2>&1 | tee -a "$out/k.txt"
done
timeout -k 10 300 python3 tools/launch_sweep.py cube4096 3840 2160 2 0 256,512,768 2>&1 | tee "$out/sweep4096.txt"
