"""bdd vs `basis` RelGraphConv layer at FB15k-237 size, h = 200: forward and forward + backward time.
    python tools/basis_bench.py      # GV_BASIS_GENERIC=1: the generic per-edge kernels instead of the relation-grouped GEMMs"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_vae_amd import sampling
from gcn_vae_amd.data import FB15K237, synthetic_kg
from gcn_vae_amd.layers import RelGraphConv
cfg = FB15K237
data = synthetic_kg(cfg['num_nodes'], cfg['num_rels'], cfg['n_train'], seed=0)
g, rel, node_norm = sampling.build_test_graph(data.num_nodes, data.num_rels, data.train)
dev = torch.device('cuda')
src, dst = g.edges()
enorm = torch.from_numpy(node_norm).to(dev)[dst.to(dev)].view(-1, 1).contiguous()
et = torch.from_numpy(rel).to(dev)
for reg, nb in (('bdd', 100), ('basis', 100), ('basis', 10)):
    torch.manual_seed(0)
    layer = RelGraphConv(200, 200, 474, reg, nb, activation=torch.relu, self_loop=True, dropout=0.0).to(dev)
    x = torch.randn(data.num_nodes, 200, device=dev, requires_grad=True)
    for it in range(3):
        y = layer(g, x, et, enorm); y.sum().backward()
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(5):
        y = layer(g, x, et, enorm)
    torch.cuda.synchronize(); t1 = time.time()
    for it in range(5):
        y = layer(g, x, et, enorm); y.sum().backward()
    torch.cuda.synchronize(); t2 = time.time()
    print(f'{reg} nb={nb}: fwd {(t1-t0)/5*1e3:.2f} ms, fwd+bwd {(t2-t1)/5*1e3:.2f} ms')
