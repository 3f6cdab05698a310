#!/bin/bash
# counters of tools/one_dispatch.py runs: tools/gpu_pmc_one.sh OUTNAME "<one_dispatch args>" ...
set -e
root=$(pwd); name=$1; shift
out=$root/gpurun_out/$name; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
for wl in "$@"; do
  tag=$(echo $wl | tr ' ' '_')
  timeout -k 10 120 python3 tools/one_dispatch.py $wl | tee -a "$out/timing.txt"
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
      --output-format csv -d "$out/pmc_$tag" -o x -- python3 tools/one_dispatch.py $wl > "$out/pmc_$tag.log" 2>&1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM \
      --output-format csv -d "$out/pmc2_$tag" -o x -- python3 tools/one_dispatch.py $wl > "$out/pmc2_$tag.log" 2>&1 || echo "pass 2 failed"
done
python3 tools/pmc_table.py "$out" | tee "$out/pmc_table.txt"
