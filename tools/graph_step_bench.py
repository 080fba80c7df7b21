#!/usr/bin/env python3
"""The mini-batch training step (kgvae/link_predict.py:200-236: sample 20 000 triplets of the FB15k-237-shaped set, split 0.5,
10 negatives, h = 200, B = 100) launched eagerly vs replayed as ONE hipGraph (gcn_vae_amd.graph_step):
    python tools/graph_step_bench.py [steps]      -> ms per step, end to end (sampling .. Adam), same process, same model."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_vae_amd.data import load_data                      # noqa: E402
from gcn_vae_amd.device_sampling import DeviceSampler       # noqa: E402
from gcn_vae_amd.encoders import KGVAE                      # noqa: E402
from gcn_vae_amd.graph_step import GraphedMiniBatchStep     # noqa: E402
from gcn_vae_amd.optim import FlatAdam                      # noqa: E402
from gcn_vae_amd.train import LinkPredict                   # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    data = load_data('FB15k-237-synthetic')
    dev = torch.device('cuda')
    torch.manual_seed(0)
    model = LinkPredict(KGVAE, data.num_nodes, 200, data.num_rels, num_bases=100, num_hidden_layers=2, dropout=0.2, use_cuda=True,
                        reg_param=0.01, kl_param=1e-5, mmd_param=1.0, k=10, n_flows=0).to(dev).train()
    opt = FlatAdam(model.parameters(), lr=1e-3, max_grad_norm=1.0)
    sm = DeviceSampler(data.train, data.num_nodes, data.num_rels, dev, seed=0)
    k, split, neg = 20000, 0.5, 10

    def eager_step():
        b = sm.sample(k, split, neg)
        opt.zero_grad()
        embed = model(b.g, b.node_id, b.edge_type, b.edge_norm)
        loss = model.get_loss(b.g, embed, b.samples, b.labels)[0]
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        eager_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eager_step()
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / steps
    step = GraphedMiniBatchStep(model, opt, sm, k, split, neg)
    for _ in range(3):
        step.body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step.body()
    torch.cuda.synchronize()
    t_static = (time.perf_counter() - t0) / steps
    step.capture(warmup=2)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / steps
    print(f'eager, synchronising sampler (dynamic shapes): {t_eager * 1e3:7.3f} ms/step')
    print(f'eager, static shapes (padded to {min(2 * k, data.num_nodes)} rows)    : {t_static * 1e3:7.3f} ms/step')
    print(f'one hipGraph replay per step                 : {t_graph * 1e3:7.3f} ms/step   (loss {float(out[0].detach()):.4f})')


if __name__ == '__main__':
    main()
