#!/bin/bash
# Round evidence on one MI355X: per-step kernel summaries, PMC traffic (separate FETCH / WRITE passes), L2 hit rates and plain
# bench lines for the bench configurations.   tools/collect_profiles.sh <outdir under gpurun_out> [configs...]
set -e -o pipefail
out=$PWD/gpurun_out/$1; shift
cfgs=${@:-c2 c4 c5}
mkdir -p "$out"
export TMPDIR=/tmp
common="--no-cpu-baseline --no-check --profile-steps 0"
for c in $cfgs; do
  echo "== $c: kernel trace"; date +%T
  rm -rf /tmp/pp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python3 bench.py --config $c --steps 10 --warmup 3 $common > "$out/bench_profiled_$c.json" 2> "$out/stderr_$c.log"
  python3 profiles/summarize_trace.py "$(find /tmp/pp -name '*kernel_trace.csv' | head -1)" 10 > "$out/per_step_summary_$c.txt"
  head -40 "$(find /tmp/pp -name '*kernel_stats.csv' | head -1)" > "$out/kernel_stats_top_$c.csv"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "== $c: pmc $ctr"; date +%T
    rm -rf /tmp/pp_$ctr
    rocprofv3 --pmc $ctr --output-format csv -d /tmp/pp_$ctr -o p -- python3 bench.py --config $c --steps 3 --warmup 2 --no-graph $common > /dev/null 2>> "$out/stderr_$c.log"
  done
  python3 profiles/summarize_pmc.py "$(find /tmp/pp_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find /tmp/pp_WRITE_SIZE -name '*counter_collection.csv' | head -1)" "$out/pmc_traffic_$c.json" "tools/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of python3 bench.py --config $c --steps 3 --warmup 2 --no-graph $common; traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB" > "$out/pmc_traffic_summary_$c.txt"
  echo "== $c: pmc TCC hit/miss"; date +%T
  rm -rf /tmp/pp_l2
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pp_l2 -o p -- python3 bench.py --config $c --steps 3 --warmup 2 --no-graph $common > /dev/null 2>> "$out/stderr_$c.log"
  python3 profiles/summarize_l2.py "$(find /tmp/pp_l2 -name '*counter_collection.csv' | head -1)" > "$out/l2_hit_rates_$c.txt"
  rm -rf /tmp/pp /tmp/pp_FETCH_SIZE /tmp/pp_WRITE_SIZE /tmp/pp_l2
done
echo "== done"; date +%T
