#!/bin/bash
# First GPU call of round 2: the new tests, the whole GPU suite, then SQ counters of the
# 40-sphere workloads (one rocprofv3 pass each).
set -e
root=$(pwd); out=$root/gpurun_out/r2a; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > "$out/wip.log" 2>&1 || { tail -30 "$out/wip.log"; exit 1; }
tail -3 "$out/wip.log"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$out/gpu_tests.log" 2>&1 || { tail -40 "$out/gpu_tests.log"; exit 1; }
tail -3 "$out/gpu_tests.log"
for wl in "default 1920 1080 1 50 0 20" "default 1280 720 64 8 0 3" "default 1920 1080 1 50 1 20" "default 1280 720 64 8 1 3"; do
  tag=$(echo $wl | tr ' ' '_')
  timeout -k 10 120 python3 tools/one_dispatch.py $wl | tee -a "$out/timing.txt"
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU \
      --output-format csv -d "$out/pmc_$tag" -o x -- python3 tools/one_dispatch.py $wl > "$out/pmc_$tag.log" 2>&1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE \
      --output-format csv -d "$out/pmc2_$tag" -o x -- python3 tools/one_dispatch.py $wl > "$out/pmc2_$tag.log" 2>&1
done
python3 tools/pmc_table.py "$out" | tee "$out/pmc_table.txt"
