#!/usr/bin/env python3
"""Which torch (ATen) kernels does the mini-batch step still launch, and from which line of the package?"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from collections import Counter
from torch.profiler import profile, ProfilerActivity
from gcn_vae_amd.data import synthetic_kg
from gcn_vae_amd.device_sampling import DeviceSampler
from gcn_vae_amd.encoders import KGVAE
from gcn_vae_amd.graph_step import GraphedMiniBatchStep
from gcn_vae_amd.optim import FlatAdam
from gcn_vae_amd.train import LinkPredict
data = synthetic_kg(14541, 237, 272115, seed=0)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = LinkPredict(KGVAE, data.num_nodes, 200, data.num_rels, num_bases=100, num_hidden_layers=2, dropout=0.2, use_cuda=True,
                    reg_param=0.01, kl_param=1e-5, mmd_param=1.0, k=10, n_flows=int(os.environ.get('FLOWS', '0'))).to(dev).train()
opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0)
sm = DeviceSampler(data.train, data.num_nodes, data.num_rels, dev, seed=0)
step = GraphedMiniBatchStep(model, opt, sm, 20000, 0.5, 10)
for _ in range(3):
    step.eager_step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
c = Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in ('view', 'as_strided', 'empty', 'detach', 'alias', 'slice', 'select', 'expand', 'reshape', 't.default', 'unsqueeze',
                                       'squeeze', 'transpose', 'permute', 'is_', 'sym_', 'stride', 'size', '_local_scalar')):
            frames = [f for f in traceback.extract_stack() if 'gcn' in f.filename and 'probes' not in f.filename]
            where = f'{os.path.basename(frames[-1].filename)}:{frames[-1].lineno} {frames[-1].line}' if frames else '?'
            c[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    step.eager_step()
torch.cuda.synchronize()
for (name, where), k in sorted(c.items(), key=lambda kv: -kv[1])[:70]:
    print(f'{k:3d} {name:32s} {where[:150]}')
