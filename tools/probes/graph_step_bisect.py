#!/usr/bin/env python3
"""Diagnostic: capture a PREFIX of the mini-batch step into a hipGraph and replay it (small configuration of
tests/test_gpu_model.py::test_train_driver_with_graph_step).   python tools/graph_step_bisect.py <stage 1..5>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd.data import load_data                      # noqa: E402
from gcn_vae_amd.device_sampling import DeviceSampler       # noqa: E402
from gcn_vae_amd.encoders import KGVAE                      # noqa: E402
from gcn_vae_amd.optim import FlatAdam                      # noqa: E402
from gcn_vae_amd.train import LinkPredict                   # noqa: E402

stage = int(sys.argv[1])
data = load_data('synthetic:400:9:3000:150:150:1')
dev = torch.device('cuda')
torch.manual_seed(0)
model = LinkPredict(KGVAE, data.num_nodes, 16, data.num_rels, num_bases=4, num_hidden_layers=2, dropout=0.2, use_cuda=True,
                    reg_param=0.01, kl_param=1e-3, mmd_param=1.0, k=4, n_flows=0).to(dev).train()
opt = FlatAdam(model.parameters(), lr=1e-2, max_grad_norm=1.0)
sm = DeviceSampler(data.train, data.num_nodes, data.num_rels, dev, seed=0)
pick = torch.zeros(200, dtype=torch.int64, device=dev)
model.encoder.mmd_index_override = pick
one = torch.ones((), device=dev)
keep = []


def body():
    b = sm.sample_static(600, 0.5, 10, mmd_pick=pick)
    keep.append(b)
    if stage == 1:
        return b.samples
    gidx = b.g.device_index(dev)
    ridx = gidx.relation_index(b.edge_type, 2 * data.num_rels)
    keep.extend([gidx, ridx])
    if stage == 2:
        return ridx.et_by_dst
    model.rows_dev = b.rows_dev
    opt.zero_grad()
    embed = model(b.g, b.node_id, b.edge_type, b.edge_norm)
    if stage == 3:
        return embed
    loss = model.get_loss(b.g, embed, b.samples, b.labels)[0]
    if stage == 4:
        return loss
    loss.backward(gradient=one.expand_as(loss))
    if stage == 5:
        return loss
    opt.step()
    return loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print(f'stage {stage}: eager ok', flush=True)
keep.clear()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    out = body()
print(f'stage {stage}: captured', flush=True)
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print(f'stage {stage}: replay {i} ok, out sum {float(out.detach().float().sum()):.4f}', flush=True)
