#!/bin/bash
# Round-4 evidence, second half (after the forward passes launch / the two-workgroup backward chain): what changed is the bf16 flow
# configurations -- their bench lines, the c3 kernel trace + PMC traffic + L2 hit rates, the kernel trace of c2 + 3 IAF blocks.
set -e -o pipefail
mkdir -p gpurun_out/r4b
for spec in "c3:--config c3" "mb_flows3_bf16:--config mb --n-flows 3 --gemm-precision bf16" "c2_flows3_bf16:--n-flows 3 --gemm-precision bf16"; do
  tag=${spec%%:*}; flags=${spec#*:}
  echo "== bench $tag"; date +%T
  timeout -k 10 700 python bench.py $flags --steps 20 --warmup 5 > gpurun_out/r4b/bench_$tag.json 2> gpurun_out/r4b/bench_$tag.err || echo "rc=$? for $tag"
done
echo "== profile c2_flows3_bf16"; date +%T
bash tools/profile_bench.sh c2_flows3_bf16 --n-flows 3 --gemm-precision bf16 > /dev/null 2>&1 || echo "profile rc=$?"
cp gpurun_out/prof_c2_flows3_bf16/per_step_summary.txt gpurun_out/r4b/per_step_summary_c2_flows3_bf16.txt
bash tools/collect_profiles.sh r4b c3
echo "== done"; date +%T
