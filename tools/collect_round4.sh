#!/bin/bash
# Round-4 evidence on one MI355X: the bench lines of every configuration (with the CPU-oracle leg and the parity leg) and the
# per-step kernel summaries of the flows / headline configurations.   tools/collect_round4.sh [benches|profiles|all]
set -e -o pipefail
what=${1:-all}
mkdir -p gpurun_out/r4
# (c5: one oracle step on 50 M edges takes minutes on the host: no CPU leg; its parity is tests/test_gpu_configs.py)
if [ "$what" = benches ] || [ "$what" = all ] || [ "$what" = c5 ]; then
for spec in "c2:" "c3:--config c3" "mb:--config mb" "mb_flows3:--config mb --n-flows 3" "mb_flows3_bf16:--config mb --n-flows 3 --gemm-precision bf16" "c2_flows3_bf16:--n-flows 3 --gemm-precision bf16" "c4:--config c4" "c5:--config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-check"; do
  tag=${spec%%:*}; flags=${spec#*:}
  if [ "$what" = c5 ] && [ "$tag" != c5 ]; then continue; fi
  echo "== bench $tag"; date +%T
  case "$flags" in *--steps*) st="";; *) st="--steps 20 --warmup 5";; esac
  timeout -k 10 700 python bench.py $flags $st > gpurun_out/r4/bench_$tag.json 2> gpurun_out/r4/bench_$tag.err || echo "rc=$? for $tag"
done
fi
if [ "$what" = profiles ] || [ "$what" = all ]; then
for spec in "c2:" "c3:--config c3" "mb:--config mb" "mb_flows3:--config mb --n-flows 3"; do
  tag=${spec%%:*}; flags=${spec#*:}
  echo "== profile $tag"; date +%T
  bash tools/profile_bench.sh $tag $flags > /dev/null 2>&1 || echo "profile rc=$? for $tag"
  cp gpurun_out/prof_$tag/per_step_summary.txt gpurun_out/r4/per_step_summary_$tag.txt
  cp gpurun_out/prof_$tag/kernel_stats.csv gpurun_out/r4/kernel_stats_top_$tag.csv
done
fi
echo "== done"; date +%T
