import os, sys, torch, random
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.argv = ['bench.py']
import bench
from torch.profiler import profile, ProfilerActivity
args = bench.parse()
from gcn_vae_amd.optim import FlatAdam
dev = torch.device('cuda')
w = bench.make_workload(0, 1, args, dev)
model = bench.build_model(w, args).to(dev).train()
g = w['g']; node_id, etype = w['node_id'].to(dev), w['rel'].to(dev); enorm, samples, labels = w['enorm'], w['samples'].to(dev), w['labels'].to(dev)
params = [p for p in model.parameters() if p.requires_grad]
opt = FlatAdam(params, lr=1e-3, max_grad_norm=1.0)
model.encoder.mmd_index_override = torch.tensor(random.sample(range(14541), 200), device=dev)
def step():
    opt.zero_grad(); e = model(g, node_id, etype, enorm); l = model.get_loss(g, e, samples, labels)[0]; l.backward(); opt.step()
for _ in range(3): step()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
from collections import Counter
c = Counter(ev.name for ev in prof.events() if ev.name.startswith('aten::'))
for n, k in c.most_common(40):
    print(f'{k:3d} {n}')
