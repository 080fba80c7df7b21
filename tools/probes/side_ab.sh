# A/B of the side stream for the fp32 MADE node's weight / bias gradients: bash tools/probes/side_ab.sh
for v in 1 0 1 0; do GV_BWD_SIDE=$v timeout -k 10 300 python bench.py --config mb --n-flows 3 --no-cpu-baseline --no-check > gpurun_out/ab_mbf_s$v.json 2>gpurun_out/ab_mbf_s$v.err; python -c "
import json; d=json.loads(open('gpurun_out/ab_mbf_s$v.json').read().strip().splitlines()[-1]); print('mb flows3 f32, side', $v, d['ms_per_step'], d.get('ms_per_step_median'))"; done
