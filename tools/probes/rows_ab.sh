# A/B of the MADE row blocks of the fp32 node: bash tools/probes/rows_ab.sh
for v in 1 2 3; do GV_MADE_F32_ROW_BLOCKS=$v timeout -k 10 300 python bench.py --config mb --n-flows 3 --no-cpu-baseline --no-check > gpurun_out/ab_mbf_b$v.json 2>gpurun_out/ab_mbf_b$v.err; python -c "
import json; d=json.loads(open('gpurun_out/ab_mbf_b$v.json').read().strip().splitlines()[-1]); print('mb flows3 f32 blocks', $v, d['ms_per_step'], d.get('ms_per_step_median'))"; done
for v in 1 2; do GV_MADE_F32_ROW_BLOCKS=$v timeout -k 10 300 python bench.py --n-flows 3 --no-cpu-baseline --no-check > gpurun_out/ab_c2ff_b$v.json 2>gpurun_out/ab_c2ff_b$v.err; python -c "
import json; d=json.loads(open('gpurun_out/ab_c2ff_b$v.json').read().strip().splitlines()[-1]); print('c2 flows3 f32 blocks', $v, d['ms_per_step'], d.get('ms_per_step_median'))"; done
