for cfg in c2 mb c4; do for v in 1 0; do GV_BWD_SIDE=$v timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-check > gpurun_out/ab_${cfg}_$v.json 2>gpurun_out/ab_${cfg}_$v.err; python -c "
import json; d=json.loads(open('gpurun_out/ab_${cfg}_$v.json').read().strip().splitlines()[-1]); print('$cfg', $v, d['ms_per_step'], d.get('ms_per_step_median'))"; done; done
