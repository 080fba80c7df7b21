#!/usr/bin/env python3
"""Diagnostic: which node KINDS does the captured mini-batch step hold?  hipMemsetAsync / hipMemcpyAsync recorded into a hipGraph
become memset / memcpy NODES, which on this stack can replay with stale parameters after other runtime work (DESIGN.md, round
2: 'Memory access fault by GPU').  Captures the step (no replay); run it under the HIP API trace and count the calls between
hipStreamBeginCapture and hipStreamEndCapture:
    rocprofv3 --hip-runtime-trace --output-format csv -d out -- python3 tools/probes/graph_nodes_probe.py <n_flows>"""
import collections
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd.data import synthetic_kg                  # noqa: E402
from gcn_vae_amd.device_sampling import DeviceSampler       # noqa: E402
from gcn_vae_amd.encoders import KGVAE                      # noqa: E402
from gcn_vae_amd.graph_step import GraphedMiniBatchStep     # noqa: E402
from gcn_vae_amd.optim import FlatAdam                      # noqa: E402
from gcn_vae_amd.train import LinkPredict                   # noqa: E402

n_flows = int(sys.argv[1]) if len(sys.argv) > 1 else 3
data = synthetic_kg(14541, 237, 272115, seed=0)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = LinkPredict(KGVAE, data.num_nodes, 200, data.num_rels, num_bases=100, num_hidden_layers=2, dropout=0.2, use_cuda=True,
                    reg_param=0.01, kl_param=1e-5, mmd_param=1.0, k=10, n_flows=n_flows).to(dev).train()
opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0)
sm = DeviceSampler(data.train, data.num_nodes, data.num_rels, dev, seed=0)
step = GraphedMiniBatchStep(model, opt, sm, 20000, 0.5, 10)
step.side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(step.side):
    for _ in range(3):
        step.body()
torch.cuda.current_stream().wait_stream(step.side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    with torch.cuda.graph(g, stream=step.side):
        step.body()
if os.environ.get('GV_PROBE_COPIES'):
    seen = 0
    for ev in prof.events():
        if ev.name in ('aten::copy_', 'aten::clone') and seen < 40:
            shapes = ev.input_shapes
            st = [f for f in (ev.stack or []) if 'gcn' in f or 'graph_step' in f or 'autograd' in f][:4]
            print('COPY', ev.name, shapes, 'thread', ev.thread, st)
            seen += 1
print('captured', n_flows)
