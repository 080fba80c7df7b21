#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, counter_collection.csv) to
per-launch HBM-side traffic of the gv:: kernels.  Units and the gfx950 correction follow
/opt/skills/guides/MI355X_MICROARCH.md §HBM: the counters are in KiB; FETCH_SIZE reads exactly half of a
wide coalesced stream's bytes on gfx950, so traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes.
    python profiles/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json] [command text]
The JSON carries "_meta": date, the command, GV_HEAD (the commit the caller names) and "k1_source_sha" -- the hash of the K1 kernel
sources and their host geometry at collection time; bench.py attaches a traffic figure to its roofline only while that hash equals
the tree's (the kernel instance a tag runs is then the one that was measured) and reports the instance's full name."""
import csv
import datetime
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K1_SOURCES = ['gcn-vae_amd/csrc/common.h', 'gcn-vae_amd/csrc/k_bdd.hip', 'gcn-vae_amd/csrc/k_phase.h', 'gcn-vae_amd/csrc/k_phase.hip',
              'gcn-vae_amd/csrc/k_stream.hip', 'gcn-vae_amd/csrc/k_lds.hip', 'gcn-vae_amd/indices.py']


def k1_source_sha(root=ROOT):
    h = hashlib.sha256()
    for rel in K1_SOURCES:
        path = os.path.join(root, rel)
        h.update(rel.encode())
        h.update(open(path, 'rb').read() if os.path.exists(path) else b'<missing>')
    return h.hexdigest()[:16]


def tag_of(name):
    m = re.search(r'gv::(k_[a-z_0-9]+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')).replace(' ', '') if m else None


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        t = tag_of(r['Kernel_Name'])
        if t:
            acc[t].append(float(r['Counter_Value']))
    return acc


def main(fetch_csv, write_csv, out=None, command=None):
    f, w = per_kernel(fetch_csv, 'FETCH_SIZE'), per_kernel(write_csv, 'WRITE_SIZE')
    res = {}
    print(f'{"kernel":44s} {"launches":>8s} {"FETCH_KiB":>11s} {"WRITE_KiB":>11s} {"traffic_MB(2F+W)":>17s}')
    for k in sorted(f):
        fv = f[k][len(f[k]) // 2:]            # later launches: caches warm, steady state
        wv = w.get(k, [0.0])[len(w.get(k, [0.0])) // 2:]
        fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
        traffic = (2 * fa + wa) * 1024
        res[k] = {'launches': len(f[k]), 'fetch_KiB': fa, 'write_KiB': wa, 'traffic_bytes': traffic}
        print(f'{k:44s} {len(f[k]):8d} {fa:11.1f} {wa:11.1f} {traffic / 1e6:17.2f}')
    if out:
        res['_meta'] = {'date': datetime.date.today().isoformat(), 'commit': os.environ.get('GV_HEAD', 'unknown'),
                        'k1_source_sha': k1_source_sha(), 'gpu': 'MI355X (gpurun box)',
                        'command': command or 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB'}
        json.dump(res, open(out, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main(*sys.argv[1:5])
