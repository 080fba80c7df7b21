#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py configuration, reduced to the per-step summary.
#   tools/profile_bench.sh <tag> [bench.py flags...]        -> gpurun_out/prof_<tag>/{per_step_summary.txt,kernel_stats.csv,bench.json}
set -e -o pipefail
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw" -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-check "$@" > "$out/bench.json" 2> "$out/stderr.log"
kt=$(find "$out/raw" -name '*kernel_trace.csv' | head -1)
ks=$(find "$out/raw" -name '*kernel_stats.csv' | head -1)
python3 profiles/summarize_trace.py "$kt" 10 > "$out/per_step_summary.txt"
head -60 "$ks" > "$out/kernel_stats.csv"
rm -rf "$out/raw"
tail -1 "$out/bench.json"
