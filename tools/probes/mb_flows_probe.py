import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from gcn_vae_amd import lib, ops
from gcn_vae_amd.data import synthetic_kg
from gcn_vae_amd.device_sampling import DeviceSampler
from gcn_vae_amd.encoders import KGVAE
from gcn_vae_amd.graph_step import GraphedMiniBatchStep
from gcn_vae_amd.optim import FlatAdam
from gcn_vae_amd.train import LinkPredict
def mark(s):
    torch.cuda.synchronize(); print('OK', s, flush=True)
mode = sys.argv[1]
data = synthetic_kg(14541, 237, 272115, seed=0)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = LinkPredict(KGVAE, data.num_nodes, 200, data.num_rels, num_bases=100, num_hidden_layers=2, dropout=0.2, use_cuda=True,
                    reg_param=0.01, kl_param=1e-5, mmd_param=1.0, k=10, n_flows=3).to(dev).train()
opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0)
sm = DeviceSampler(data.train, data.num_nodes, data.num_rels, dev, seed=0)
step = GraphedMiniBatchStep(model, opt, sm, 20000, 0.5, 10)
mark('built')
if 'all' in mode:
    step.capture(warmup=3); mark('captured')
    for _ in range(3): step()
    mark('replayed')
b = sm.sample(20000, 0.5, 10); mark('sampled n=%d' % b.node_id.shape[0])
enc = model.encoder
n = b.node_id.shape[0]
if mode != 'nooverride':
    enc.eps_override = torch.randn(n, 200, device=dev); enc.mmd_eps_override = torch.randn(200, 200, device=dev)
    enc.mmd_index_override = torch.randperm(n, device=dev)[:200]
    enc.rconv_layer_1.keep_mask_override = (torch.rand(n, 200, device=dev) > 0.2).to(torch.uint8)
    enc.rconv_layer_2.keep_mask_override = (torch.rand(n, 400, device=dev) > 0.2).to(torch.uint8)
opt.flat_g.zero_(); opt._mark_fresh()
import contextlib
ctxm = torch.cuda.stream(step.side) if 'side' in mode else contextlib.nullcontext()
if 'side' in mode:
    step.side.wait_stream(torch.cuda.current_stream())
with ctxm:
    embed = model(b.g, b.node_id, b.edge_type, b.edge_norm); mark('forward')
    loss = model.get_loss(b.g, embed, b.samples, b.labels)[0]; mark('loss %.4f' % float(loss.detach()))
    if 'nobwd' not in mode:
        loss.backward(); mark('backward')
if 'side' in mode:
    torch.cuda.current_stream().wait_stream(step.side)
del embed, loss
if 'all' in mode:
    enc.eps_override = enc.mmd_eps_override = enc.mmd_index_override = None
    enc.rconv_layer_1.keep_mask_override = enc.rconv_layer_2.keep_mask_override = None
    opt.flat_g.zero_()
    for _ in range(5): step()
    mark('replayed after eager')
