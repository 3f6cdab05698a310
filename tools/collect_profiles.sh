#!/bin/bash
# Run on the GPU box (through gpurun): kernel statistics and HBM traffic counters of `python3 bench.py`.
# Four separate rocprofv3 runs (timing; FETCH_SIZE; WRITE_SIZE; SQ instruction counts), each under its own time limit.
# Results land in gpurun_out/prof/<tag>/; tools/summarise_profiles.py turns them into profiles/.
#   usage: tools/collect_profiles.sh TAG [bench.py arguments]
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- \
    python3 bench.py --cpu-seconds 0 "$@" > "$out/bench_stats.log" 2>&1
grep '^{' "$out/bench_stats.log" > "$out/bench_line.json"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o bench -- \
    python3 bench.py --cpu-seconds 0 "$@" > "$out/bench_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o bench -- \
    python3 bench.py --cpu-seconds 0 "$@" > "$out/bench_write.log" 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES \
    --output-format csv -d "$out/pmc_sq" -o bench -- \
    python3 bench.py --cpu-seconds 0 "$@" > "$out/bench_sq.log" 2>&1
find "$out" -name "*.csv" | sort
