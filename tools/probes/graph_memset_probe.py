#!/usr/bin/env python3
"""Diagnostic: capture sample_static with only the first K native calls issued (the rest skipped), replay 4 times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import lib                                 # noqa: E402
from gcn_vae_amd.data import load_data                      # noqa: E402
from gcn_vae_amd.device_sampling import DeviceSampler       # noqa: E402
import gcn_vae_amd.device_sampling as ds                    # noqa: E402

K = int(sys.argv[1])
data = load_data('synthetic:400:9:3000:150:150:1')
torch.manual_seed(0)
sm = DeviceSampler(data.train, data.num_nodes, data.num_rels, 'cuda', seed=0)
pick = torch.zeros(200, dtype=torch.int64, device='cuda')
orig = lib.call
state = {'n': 0, 'limit': 10 ** 9}


def limited(name, *a, **k):
    state['n'] += 1
    if state['n'] > state['limit']:
        return 0
    return orig(name, *a, **k)


ds.lib.call = limited
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        state['n'] = 0
        sm.sample_static(600, 0.5, 10, mmd_pick=pick)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
state['n'], state['limit'] = 0, K
with torch.cuda.graph(g, stream=side):
    b = sm.sample_static(600, 0.5, 10, mmd_pick=pick)
mode = sys.argv[2] if len(sys.argv) > 2 else ''
for i in range(4):
    g.replay()
    torch.cuda.synchronize()
    if mode == 'read':
        print('  sum', float(b.samples.detach().float().sum()), 'count', int(b.rows_dev.item()), flush=True)
    if mode == 'count':
        print('  count', int(b.rows_dev.item()), 'uniq max', int(b.node_id.max()), 'pick max', int(pick.max()), flush=True)
print(f'K={K}: 4 replays ok', flush=True)
