"""Launched by tests/test_gpu_zz_dist.py: the DESTINATION-ROW partition of the HIP path (distributed.RowPartition,
ops.rel_graph_conv_rows, AllGatherRows, LinkPredict._get_loss_rows) against a single-process HIP run on the whole graph.

  * under torch.distributed.run with TWO ranks sharing cuda:0 (backend gloo -- RCCL refuses two ranks on one device);
  * or as a plain process (world 1), where the row path degenerates to local copies.

Every rank owns a block of node rows (dealt by in-degree) and the edges that end in them, and a slice of the triplets.
Checked: sum over ranks of the loss shares == the full loss (with L = mean_p(pred_p + mmd) + reg + kl; the reference run
scores ALL triplets, so pred is compared through equal-sized slices), summed gradients == the full gradients."""
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
from gcn_vae_amd.train import LinkPredict      # noqa: E402


def main():
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        dist.init_process_group('gloo')
    rank = dist.get_rank() if world > 1 else 0
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    n, n_rel, h, nb = 1501, 60, 200, 100       # odd node count: one rank's slot ends in a padding row
    data = synthetic_kg(n, n_rel, 9000, seed=0)
    g_full, rel, node_norm = sampling.build_test_graph(n, n_rel, data.train)
    src, dst = (t.numpy() for t in g_full.edges())
    torch.manual_seed(0)
    net = LinkPredict(KGVAE, n, h, n_rel, num_bases=nb, num_hidden_layers=2, dropout=0.2, use_cuda=True, reg_param=0.01,
                      kl_param=1e-3, mmd_param=1.0, k=10, n_flows=int(os.environ.get('GV_WORKER_FLOWS', '0'))).to(dev).train()
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

    # ---- single-process run on the whole graph, all triplets
    ref = copy.deepcopy(net)
    e = ref.encoder
    e.eps_override, e.mmd_eps_override, e.mmd_index_override = eps, eps_prior, pick
    e.rconv_layer_1.keep_mask_override, e.rconv_layer_2.keep_mask_override = keep1, keep2
    norm_full = torch.from_numpy(node_norm).to(dev)
    enorm_full = norm_full[torch.from_numpy(dst).to(dev)].view(-1, 1).contiguous()
    ref.zero_grad()
    embed = ref(g_full, node_id, torch.from_numpy(rel).to(dev), enorm_full)
    loss_ref = ref.get_loss(g_full, embed, torch.from_numpy(samples).to(dev), torch.from_numpy(labels).to(dev))[0]
    loss_ref.backward()
    grads_ref = {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()

    # ---- this rank's row block, relabelled to positions, and its triplet slice
    in_deg = np.bincount(dst, minlength=n)
    part = gdist.make_row_partition(in_deg, world, rank)
    assert part.real_rows == n and part.total_rows >= n
    if world == 2:
        assert part.total_rows == n + 1, 'expected exactly one padding row'
    pos_of_node = torch.from_numpy(part.pos_of_node).to(dev)
    node_of_pos = torch.from_numpy(part.node_of_pos).to(dev)
    nop = node_of_pos.clamp(min=0)
    g_loc, et_loc, enorm_loc = gdist.build_row_block(part, part.pos_of_node, src, dst, rel, node_norm, dev)
    n_edges = torch.tensor([g_loc.number_of_edges()], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(n_edges)
    assert int(n_edges) == len(src), f'row blocks hold {int(n_edges)} of {len(src)} edges'
    T = len(samples)
    sl = slice(rank * T // world, (rank + 1) * T // world)
    trip = torch.from_numpy(samples[sl]).to(dev)
    trip_pos = torch.stack([pos_of_node[trip[:, 0]], trip[:, 1], pos_of_node[trip[:, 2]]], 1).contiguous()
    e = net.encoder
    e.row_part = part
    # overrides are per node: move them to position order (padding rows: anything -- they are never read)
    e.eps_override, e.mmd_eps_override, e.mmd_index_override = eps[nop].contiguous(), eps_prior, pos_of_node[pick]
    e.rconv_layer_1.keep_mask_override, e.rconv_layer_2.keep_mask_override = keep1[nop].contiguous(), keep2[nop].contiguous()
    net.zero_grad()
    z_all = net(g_loc, nop.view(-1, 1), et_loc, enorm_loc)
    assert z_all.shape[0] == part.total_rows
    pad = node_of_pos < 0
    if bool(pad.any()):
        assert float(z_all.detach()[pad].abs().max()) == 0.0, 'padding rows of z must be zero'
    zerr = float((z_all.detach()[pos_of_node] - embed.detach()).abs().max()) / float(embed.detach().abs().max())
    loss_share = net.get_loss(g_loc, z_all, trip_pos, torch.from_numpy(labels[sl]).to(dev))[0]
    loss_share.backward()
    params = [p for p in net.parameters() if p.requires_grad]
    total = loss_share.detach().reshape(1).clone()
    if world > 1:
        for p in params:
            if p.grad is not None:
                dist.all_reduce(p.grad)
        dist.all_reduce(total)
    torch.cuda.synchronize()

    def rel_err(a, b):
        return float((a - b).abs().max()) / max(1e-12, float(b.abs().max()))

    # the reference's BCE is a mean over ALL triplets = the mean of the (equal-sized) slices' means
    errs = {'z': zerr, 'loss': abs(float(total) - float(loss_ref)) / max(1e-12, abs(float(loss_ref)))}
    for k, p in net.named_parameters():
        if k in grads_ref:
            errs[k] = rel_err(p.grad, grads_ref[k])
    worst = max(errs, key=errs.get)
    print(f'rank {rank}/{world}: rows {part.own_rows} (slot {part.slot_rows}) edges {g_loc.number_of_edges()}/{len(src)}  '
          f'loss sum {float(total):.6f} ref {float(loss_ref):.6f}  z err {zerr:.2e}  worst rel err {errs[worst]:.2e} ({worst})',
          flush=True)
    ok = errs[worst] < 2e-4
    if world > 1:
        flag = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = float(flag) == 1.0
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
