#!/bin/bash
# Development aid: duration of the helper kernels of a workload.  usage: tools/gpu_order_prof.sh TAG lib.so "workload args"
set -e
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$1; mkdir -p $out
RT_AMD_LIB=$root/$2 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o t -- python3 $root/tools/one_dispatch.py $3 > $out/log.txt 2>&1
f=$(find $out/p -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
