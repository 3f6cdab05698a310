#!/bin/bash
# Run on the GPU box (through gpurun): kernel statistics and hardware counters of one bench.py workload.
# Separate rocprofv3 runs (timing; FETCH_SIZE; WRITE_SIZE; SQ + GRBM counters), each under its own time limit.
# Results land in gpurun_out/prof/<tag>/<workload>/; tools/summarise_profiles.py turns them into profiles/.
#   usage: tools/collect_profiles.sh TAG WORKLOAD [more bench.py arguments]   (with --kernel K the results go to <WORKLOAD>_K)
set -e
tag=$1; wl=$2; shift 2
root=$(pwd)
# a run with "--kernel K" lands in <workload>_K
label=$wl
prev=""
for a in "$@"; do
  if [ "$prev" = "--kernel" ]; then label=${wl}_$a; fi
  prev=$a
done
out=$root/gpurun_out/prof/$tag/$label
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
args="--workload $wl --cpu-seconds 0 $*"
echo "$args" > "$out/bench_args.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- \
    python3 bench.py $args > "$out/bench_stats.log" 2>&1
grep '^{' "$out/bench_stats.log" > "$out/bench_line.json"
# RT_COLLECT_LINES_ONLY=1: only the profiled and the plain bench line again (second pass: the lines then quote the
# counters that the first pass put into profiles/<tag>_pmc.json)
if [ -n "$RT_COLLECT_LINES_ONLY" ]; then
  timeout -k 10 300 python3 bench.py $args > "$out/bench_plain.log" 2>&1
  grep '^{' "$out/bench_plain.log" > "$out/bench_line_plain.json"
  echo "collected lines of $label"
  exit 0
fi
timeout -s KILL 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o bench -- \
    python3 bench.py $args > "$out/bench_fetch.log" 2>&1
timeout -s KILL 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o bench -- \
    python3 bench.py $args > "$out/bench_write.log" 2>&1
timeout -s KILL 180 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES \
    SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/pmc_sq" -o bench -- \
    python3 bench.py $args > "$out/bench_sq.log" 2>&1
timeout -s KILL 180 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA \
    SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS \
    --output-format csv -d "$out/pmc_wait" -o bench -- \
    python3 bench.py $args > "$out/bench_wait.log" 2>&1 || echo "wait pass failed (optional)"
# the unprofiled line of the same command, for the record next to the profiled one
timeout -k 10 300 python3 bench.py $args > "$out/bench_plain.log" 2>&1
grep '^{' "$out/bench_plain.log" > "$out/bench_line_plain.json"
# clock and power under load: the same command with enough steps to last some seconds, the device sampled meanwhile
# (rocm-smi, read-only: shader clock and socket power), to set the counter-derived clock beside what the part reports
steps=600; case "$wl" in interactive) steps=60000;; default720p_d8) steps=4000;; default1080p) steps=2500;; esac
timeout -k 10 200 python3 bench.py $args --steps $steps > "$out/bench_long.log" 2>&1 &
pid=$!
: > "$out/smi_samples.jsonl"
while kill -0 $pid 2>/dev/null; do
  timeout 5 rocm-smi -d 0 -c -P --json >> "$out/smi_samples.jsonl" 2>/dev/null || true
  echo >> "$out/smi_samples.jsonl"
  sleep 0.4
done
wait $pid || echo "long run failed (optional)"
grep '^{' "$out/bench_long.log" > "$out/bench_line_long.json" || true
echo "collected $label"
