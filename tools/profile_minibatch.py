#!/usr/bin/env python3
"""Where a mini-batch training step of the driver spends its wall time (host sampling, H2D, index build, GPU step)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_vae_amd import ops, sampling
from gcn_vae_amd.data import load_data
from gcn_vae_amd.encoders import KGVAE
from gcn_vae_amd.optim import FlatAdam
from gcn_vae_amd.train import LinkPredict

data = load_data('FB15k-237-synthetic')
dev = torch.device('cuda')
torch.manual_seed(0)
model = LinkPredict(KGVAE, data.num_nodes, 200, data.num_rels, num_bases=100, num_hidden_layers=2, dropout=0.2, use_cuda=True,
                    reg_param=0.01, kl_param=1e-5, mmd_param=1.0, k=10, n_flows=0).to(dev)
opt = FlatAdam(model.parameters(), lr=1e-3, max_grad_norm=1.0)
t0 = time.time(); adj, deg = sampling.get_adj_and_degrees(data.num_nodes, data.train); print(f'get_adj_and_degrees {time.time()-t0:.2f} s (once)')
from gcn_vae_amd.device_sampling import DeviceSampler
USE_DEV = len(sys.argv) > 1 and sys.argv[1] == 'device'
sm = DeviceSampler(data.train, data.num_nodes, data.num_rels, dev, seed=0)
acc = {}
def tick(name, t):
    torch.cuda.synchronize(); now = time.time(); acc[name] = acc.get(name, 0.0) + now - t; return now
for it in range(25):
    if it == 5: acc.clear()
    t = time.time()
    if USE_DEV:
        b = sm.sample(20000, 0.5, 10)
        g, node_id, etype, enorm, batch, labels = b.g, b.node_id, b.edge_type, b.edge_norm, b.samples, b.labels
        t = tick('1 device sampling', t)
    else:
        g, node_id, etype, node_norm, batch, labels = sampling.generate_sampled_graph_and_labels(
            data.train, 20000, 0.5, data.num_rels, adj, deg, 10, 'uniform')
        t = tick('1 host sampling (numpy)', t)
        node_id = torch.from_numpy(node_id).view(-1, 1).long().to(dev); etype = torch.from_numpy(etype).to(dev)
        enorm = sampling.node_norm_to_edge_norm(g, torch.from_numpy(node_norm).view(-1, 1)).to(dev)
        batch, labels = torch.from_numpy(batch).to(dev), torch.from_numpy(labels).to(dev)
        t = tick('2 H2D copies', t)
    gidx = g.device_index(dev); ridx = gidx.relation_index(etype, 2 * data.num_rels)
    t = tick('3 graph + relation index build', t)
    tidx = model.triplet_index(torch.empty(len(g), 1, device=dev), batch)
    t = tick('4 triplet index build', t)
    opt.zero_grad(); embed = model(g, node_id, etype, enorm); loss = model.get_loss(g, embed, batch, labels)[0]
    t = tick('5 forward + loss', t)
    loss.backward()
    t = tick('6 backward', t)
    opt.step()
    t = tick('7 clip + Adam', t)
n = 20
ms = torch.cuda.memory_stats()
print('device mallocs (segments) total:', ms.get('num_device_alloc'), 'frees:', ms.get('num_device_free'), 'retries:', ms.get('num_alloc_retries'), 'reserved GB:', round(torch.cuda.memory_reserved()/2**30, 2))
for k in sorted(acc): print(f'{k:36s} {acc[k] / n * 1e3:8.2f} ms/step')
print(f'{"total":36s} {sum(acc.values()) / n * 1e3:8.2f} ms/step   (E={g.number_of_edges()}, N={len(g)}, T={len(batch)})')
