# A/B of the MADE row blocks: bash tools/probes/rows_ab.sh
for v in 2 3 4; do GV_MADE_ROW_BLOCKS=$v timeout -k 10 300 python bench.py --config c3 --no-cpu-baseline --no-check > gpurun_out/ab_c3_b$v.json 2>gpurun_out/ab_c3_b$v.err; python -c "
import json; d=json.loads(open('gpurun_out/ab_c3_b$v.json').read().strip().splitlines()[-1]); print('c3 blocks', $v, d['ms_per_step'], d.get('ms_per_step_median'))"; done
for v in 1 2 3; do GV_MADE_ROW_BLOCKS=$v GV_MADE_ROW_BLOCKS_MIN_TILES=1 timeout -k 10 300 python bench.py --n-flows 3 --gemm-precision bf16 --no-cpu-baseline --no-check > gpurun_out/ab_c2f_b$v.json 2>gpurun_out/ab_c2f_b$v.err; python -c "
import json; d=json.loads(open('gpurun_out/ab_c2f_b$v.json').read().strip().splitlines()[-1]); print('c2f blocks', $v, d['ms_per_step'], d.get('ms_per_step_median'))"; done
