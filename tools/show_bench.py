"""Print the headline fields of bench.py JSON lines:  python tools/show_bench.py file.json [...]"""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print(path.split('/')[-1], round(d['ms_per_step'], 3), 'ms', round(d['value'] / 1e6, 1), 'Medges/s', r.get('kernel'), r.get('bound'),
          'frac', r.get('frac'), 'traffic/alg', r.get('traffic_over_algorithmic'))
    for k, v in (d.get('roofline_detail') or {}).items():
        print('   ', k, v['avg_us'], 'us', v.get('frac'), v.get('frac_of_hbm_peak'))
