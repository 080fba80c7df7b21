#!/usr/bin/env python3
"""Probe: the one-hipGraph mini-batch step at bench size, replayed N times with a device-to-host read after every replay (what
train.py --graph-step does: loss.item()).  Exits non-zero on any runtime fault (the process aborts).
    python tools/probes/graph_replay_reads.py [replays]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd.data import load_data                      # noqa: E402
from gcn_vae_amd.device_sampling import DeviceSampler       # noqa: E402
from gcn_vae_amd.encoders import KGVAE                      # noqa: E402
from gcn_vae_amd.graph_step import GraphedMiniBatchStep     # noqa: E402
from gcn_vae_amd.optim import FlatAdam                      # noqa: E402
from gcn_vae_amd.train import LinkPredict                   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
data = load_data('FB15k-237-synthetic')
torch.manual_seed(0)
model = LinkPredict(KGVAE, data.num_nodes, 200, data.num_rels, num_bases=100, num_hidden_layers=2, dropout=0.2, use_cuda=True,
                    reg_param=0.01, kl_param=1e-5, mmd_param=1.0, k=10, n_flows=0).cuda().train()
opt = FlatAdam(model.parameters(), lr=1e-3, max_grad_norm=1.0)
sm = DeviceSampler(data.train, data.num_nodes, data.num_rels, 'cuda', seed=0)
step = GraphedMiniBatchStep(model, opt, sm, 20000, 0.5, 10).capture()
acc = 0.0
for i in range(n):
    loss, pred, kl, mmd = step()
    acc += loss.item() + kl.item()                 # device-to-host copies between replays
    if i % 50 == 0:
        junk = torch.randn(1000 + i, device='cuda').sum().item()      # other allocations / kernels / copies in between
        print(f'replay {i}: loss {loss.item():.4f}', flush=True)
print(f'{n} replays with host reads in between: ok (mean loss+kl {acc / n:.4f})')
