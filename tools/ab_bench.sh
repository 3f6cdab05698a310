#!/bin/bash
# A/B timing of library variants on one GPU: tools/ab_bench.sh "<quick_bench args>" lib1.so lib2.so ...
# (each run under its own time limit; stops at the first failure)
args=$1; shift
for lib in "$@"; do
    echo "=== $lib"
    RT_AMD_LIB=$lib timeout -k 10 150 python tools/quick_bench.py $args || exit 1
done
