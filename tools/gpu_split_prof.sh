#!/bin/bash
# Development aid: kernel durations of one workload with and without the frame split (variants/libknobs.so).
set -e
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$1; mkdir -p $out
for den in 0 4; do
  RT_DEBUG_STREAM_SPLIT_DEN=$den RT_AMD_LIB=$root/variants/libknobs.so timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/den$den -o t -- python3 $root/tools/one_dispatch.py ${RT_WL:-cube200 3840 2160 8 50 0 16} > $out/den$den.log 2>&1
  f=$(find $out/den$den -name "*kernel_stats.csv" | head -1); cp $f $out/den${den}_kernel_stats.csv
done
