#!/bin/bash
# Round-5 evidence on one MI355X for ONE configuration: bench line, in-step kernel trace, SERIALISED kernel trace (side streams
# off: every kernel alone -- the "alone" fractions of DESIGN.md), PMC traffic (FETCH / WRITE, separate passes) and L2 hit rates.
#   GV_HEAD=<commit> [SKIP_PMC=1] tools/collect_round5.sh <config> [bench flags for the bench line]
set -e -o pipefail
c=$1; shift
out=$PWD/gpurun_out/r5
mkdir -p "$out"
export TMPDIR=/tmp
common="--no-cpu-baseline --no-check --profile-steps 0"
steps="--steps 10 --warmup 3"; psteps="--steps 3 --warmup 2"
if [ "$c" = c5 ]; then steps="--steps 3 --warmup 1"; psteps="--steps 2 --warmup 1"; fi
echo "== $c: bench line"; date +%T
timeout -k 10 900 python bench.py --config $c "$@" > "$out/bench_$c.json" 2> "$out/bench_$c.err" || echo "bench rc=$?"
for mode in instep serial; do
  echo "== $c: kernel trace ($mode)"; date +%T
  rm -rf /tmp/pp
  if [ $mode = serial ]; then export GV_BWD_SIDE=0 GV_RGCN_BWD_SIDE=0 GV_MADE_PREPARE=0; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python3 bench.py --config $c $steps $common > /dev/null 2> "$out/stderr_$c.log"
  n=10; if [ "$c" = c5 ]; then n=3; fi
  sfx=""; if [ $mode = serial ]; then sfx="_serial"; fi
  python3 profiles/summarize_trace.py "$(find /tmp/pp -name '*kernel_trace.csv' | head -1)" $n > "$out/per_step_summary_$c$sfx.txt"
  if [ $mode = instep ]; then head -40 "$(find /tmp/pp -name '*kernel_stats.csv' | head -1)" > "$out/kernel_stats_top_$c.csv"; fi
  unset GV_BWD_SIDE GV_RGCN_BWD_SIDE GV_MADE_PREPARE
done
if [ "${SKIP_PMC:-0}" = 1 ]; then echo "== done (no counter passes: SKIP_PMC=1)"; rm -rf /tmp/pp; exit 0; fi
for ctr in FETCH_SIZE WRITE_SIZE; do
  echo "== $c: pmc $ctr"; date +%T
  rm -rf /tmp/pp_$ctr
  rocprofv3 --pmc $ctr --output-format csv -d /tmp/pp_$ctr -o p -- python3 bench.py --config $c $psteps --no-graph $common > /dev/null 2>> "$out/stderr_$c.log"
done
python3 profiles/summarize_pmc.py "$(find /tmp/pp_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find /tmp/pp_WRITE_SIZE -name '*counter_collection.csv' | head -1)" "$out/pmc_traffic_$c.json" "tools/collect_round5.sh $c: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of python3 bench.py --config $c $psteps --no-graph $common; traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB" > "$out/pmc_traffic_summary_$c.txt"
echo "== $c: pmc TCC hit/miss"; date +%T
rm -rf /tmp/pp_l2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pp_l2 -o p -- python3 bench.py --config $c $psteps --no-graph $common > /dev/null 2>> "$out/stderr_$c.log"
python3 profiles/summarize_l2.py "$(find /tmp/pp_l2 -name '*counter_collection.csv' | head -1)" > "$out/l2_hit_rates_$c.txt"
rm -rf /tmp/pp /tmp/pp_FETCH_SIZE /tmp/pp_WRITE_SIZE /tmp/pp_l2
echo "== done"; date +%T
