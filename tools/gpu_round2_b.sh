#!/bin/bash
set -e
root=$(pwd); out=$root/gpurun_out/r2b; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > "$out/zz_alone.log" 2>&1 || { tail -30 "$out/zz_alone.log"; exit 1; }
tail -2 "$out/zz_alone.log"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$out/gpu_tests.log" 2>&1 || { tail -40 "$out/gpu_tests.log"; exit 1; }
tail -3 "$out/gpu_tests.log"
for e in 0 4 6 7 8; do
  echo "== static eighths $e" | tee -a "$out/static.txt"
  RT_DEBUG_STATIC_EIGHTHS=$e timeout -k 10 120 python3 tools/one_dispatch.py default 1920 1080 1 50 0 200 | tee -a "$out/static.txt"
  RT_DEBUG_STATIC_EIGHTHS=$e timeout -k 10 120 python3 tools/one_dispatch.py default 1280 720 64 8 0 5 | tee -a "$out/static.txt"
  RT_DEBUG_STATIC_EIGHTHS=$e timeout -k 10 120 python3 tools/one_dispatch.py rtiow 3840 2160 8 50 0 3 | tee -a "$out/static.txt"
done
RT_DEBUG_STATIC_EIGHTHS=7 timeout -k 10 200 python3 tools/frame_latency.py 1920 1080 1 50 | tee "$out/lat7.txt"
