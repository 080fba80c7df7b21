"""Launched by tests/test_gpu_zz_dist.py under torch.distributed.run with TWO ranks that share cuda:0 (backend gloo, which
moves CUDA tensors through the host -- RCCL refuses two ranks on one device).  Runs the edge-sharded HIP path
(reduce hooks on both R-GCN layers, averaged flat gradients) and checks it against a single-process HIP run on the
union graph:  mean over ranks of the rank losses == the full loss,  averaged gradients == the full gradients."""
import copy
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gcn_vae_amd import distributed as gdist   # noqa: E402
from gcn_vae_amd import sampling               # noqa: E402
from gcn_vae_amd.data import synthetic_kg      # noqa: E402
from gcn_vae_amd.encoders import KGVAE         # noqa: E402
from gcn_vae_amd.graph import KGraph           # noqa: E402
from gcn_vae_amd.train import LinkPredict      # noqa: E402


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    # num_bases is clamped to the number of relation types: keep R >= 100.  GV_WORKER_HIDDEN=500: BASELINE configs[3]'s width
    n, n_rel, h, nb = 1500, 60, int(os.environ.get('GV_WORKER_HIDDEN', '200')), 100
    data = synthetic_kg(n, n_rel, 9000, seed=0)
    g_full, rel, node_norm = sampling.build_test_graph(n, n_rel, data.train)
    src, dst = (t.numpy() for t in g_full.edges())
    R = 2 * n_rel
    torch.manual_seed(0)
    net = LinkPredict(KGVAE, n, h, n_rel, num_bases=nb, num_hidden_layers=2, dropout=0.2, use_cuda=True, reg_param=0.01,
                      kl_param=1e-3, mmd_param=1.0, k=10, n_flows=0).to(dev).train()
    gen = torch.Generator().manual_seed(1)
    eps, eps_prior = torch.randn(n, h, generator=gen).to(dev), torch.randn(200, h, generator=gen).to(dev)
    keep1 = (torch.rand(n, h, generator=gen) > 0.2).to(torch.uint8).to(dev)
    keep2 = (torch.rand(n, 2 * h, generator=gen) > 0.2).to(torch.uint8).to(dev)
    random.seed(2)
    pick = torch.tensor(random.sample(range(n), 200), device=dev)
    np.random.seed(3)
    pos = data.train[np.random.choice(len(data.train), 1000, replace=False)]
    samples, labels = sampling.negative_sampling(pos, n, 3)                     # T = 4000, even
    perm = np.random.permutation(len(samples))
    samples, labels = samples[perm], labels[perm]
    node_id = torch.arange(n, device=dev).view(-1, 1)

    def configure(m):
        e = m.encoder
        e.eps_override, e.mmd_eps_override, e.mmd_index_override = eps, eps_prior, pick
        e.rconv_layer_1.keep_mask_override, e.rconv_layer_2.keep_mask_override = keep1, keep2

    def step(m, graph, etype, enorm, trip, lab, hook):
        m.zero_grad()
        m.encoder.rconv_layer_1.reduce_hook = m.encoder.rconv_layer_2.reduce_hook = hook
        embed = m(graph, node_id, etype, enorm)
        loss = m.get_loss(graph, embed, trip, lab)[0]
        loss.backward()
        return loss.detach().reshape(()).clone()

    # ---- single-process run on the union graph, all triplets
    ref = copy.deepcopy(net)
    configure(ref)
    norm_full = torch.from_numpy(node_norm).to(dev)
    enorm_full = norm_full[torch.from_numpy(dst).to(dev)].view(-1, 1).contiguous()
    loss_ref = step(ref, g_full, torch.from_numpy(rel).to(dev), enorm_full, torch.from_numpy(samples).to(dev),
                    torch.from_numpy(labels).to(dev), None)
    grads_ref = {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    print(f'rank {rank}: single-process reference done, loss {float(loss_ref):.6f}', flush=True)
    if os.environ.get('GV_WORKER_REF_ONLY') == '1':
        dist.destroy_process_group()
        return

    # ---- this rank's edge block (whole relations) and triplet slice
    ids, (lo, hi) = gdist.shard_edges_by_relation(rel, R, world, rank)
    g_loc = KGraph()
    g_loc.add_nodes(n)
    g_loc.add_edges(src[ids], dst[ids])
    norm_union = gdist.global_in_degree_norm(dst[ids], n, device=dev)
    assert torch.allclose(norm_union, norm_full), 'union in-degree norm differs from the full graph norm'
    enorm_loc = norm_union[torch.from_numpy(dst[ids]).to(dev)].view(-1, 1).contiguous()
    T = len(samples)
    sl = slice(rank * T // world, (rank + 1) * T // world)
    configure(net)
    # the gradient arena in backward-completion order; its tail (layer 2, decoder, prior) is all-reduced under layer 1's backward
    from gcn_vae_amd.optim import FlatAdam
    opt = FlatAdam(gdist.arena_order(net), lr=1e-3, max_grad_norm=1.0)
    red = gdist.BucketedArenaReduce(opt.flat_g, opt.offsets)
    net.encoder.grad_reducer = red
    opt.zero_grad()
    net.zero_grad = lambda *a, **k: None          # step() clears through the module: the arena is already clean
    loss_rank = step(net, g_loc, torch.from_numpy(rel[ids]).to(dev), enorm_loc, torch.from_numpy(samples[sl]).to(dev),
                     torch.from_numpy(labels[sl]).to(dev), gdist.make_reduce_hook(async_op=True))
    launched = list(red.log)
    l2_off = opt.offsets[next(net.encoder.rconv_layer_2.parameters())]
    assert launched == [(l2_off, opt.flat_g.numel())], f'the arena tail was not reduced under backward: {launched}'
    red.finish()
    mean_loss = loss_rank.clone()
    dist.all_reduce(mean_loss)
    mean_loss /= world
    torch.cuda.synchronize()

    def rel_err(a, b):
        return float((a - b).abs().max()) / max(1e-12, float(b.abs().max()))

    errs = {'loss': abs(float(mean_loss) - float(loss_ref)) / max(1e-12, abs(float(loss_ref)))}
    for k, p in net.named_parameters():
        if k in grads_ref:
            errs[k] = rel_err(p.grad, grads_ref[k])
    worst = max(errs, key=errs.get)
    print(f'rank {rank}: relations [{lo},{hi}) edges {len(ids)}/{len(src)}  loss_rank {float(loss_rank):.6f} '
          f'mean {float(mean_loss):.6f} ref {float(loss_ref):.6f}  worst rel err {errs[worst]:.2e} ({worst})', flush=True)
    ok = errs[worst] < 2e-4 and 0 < len(ids) < len(src)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if float(flag) == 1.0 else 1)


if __name__ == '__main__':
    main()
