"""Does the ORDER of the work items matter to the K1 launches of the default configuration?  (All ~4 300 workgroups of such a launch are
resident within two rounds; a CU is dealt every 32nd slot of its XCD.)  The same item list as built, largest items first, smallest
first and shuffled: us per launch, forward 2x2 / 2x4 (lane-packed weights) and backward-x 4x2.
    python tools/probes/k1_item_order.py"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from microbench import timeit  # noqa: E402

from gcn_vae_amd import ops, sampling  # noqa: E402
from gcn_vae_amd.data import FB15K237, synthetic_kg  # noqa: E402

cfg = FB15K237
data = synthetic_kg(cfg['num_nodes'], cfg['num_rels'], cfg['n_train'], seed=0)
g, rel, node_norm = sampling.build_test_graph(data.num_nodes, data.num_rels, data.train)
src, dst = g.edges()
N, E, R = data.num_nodes, src.numel(), 2 * data.num_rels
gidx = ops.GraphIndex(src.cuda(), dst.cuda(), N)
ridx = ops.RelationIndex(gidx, torch.from_numpy(rel).cuda(), R)
norm = torch.from_numpy(node_norm).cuda()[dst.cuda()].contiguous()
nb = 100


def reordered(seg, how):
    s = copy.copy(seg)
    n = seg.n_items
    it = seg.items[:n]
    size = (it[:, 2] - it[:, 1]).long()
    valid = it[:, 0] >= 0
    size = torch.where(valid, size, torch.full_like(size, -1))
    if how == 'as built':
        return s
    if how == 'largest first':
        order = torch.argsort(size, descending=True, stable=True)
    elif how == 'smallest first':
        order = torch.argsort(torch.where(valid, size, torch.full_like(size, 1 << 40)), stable=True)
    else:
        gen = torch.Generator(device='cuda').manual_seed(1)
        order = torch.randperm(n, device='cuda', generator=gen)
    items = seg.items.clone()
    items[:n] = it[order]
    s.items = items
    return s


print(f'N={N} E={E}: forward items {gidx.by_dst.seg.n_items}, backward items {gidx.by_src.seg.n_items}', flush=True)
for (fin, fout) in ((200, 200), (200, 400)):
    si, so = fin // nb, fout // nb
    x = torch.randn(N, fin, device='cuda')
    gg = torch.randn(N, fout, device='cuda')
    w = torch.randn(R, nb * si * so, device='cuda')
    wp = ops.pack_weight(w, nb, si, so)
    wpt = ops.pack_weight(w, nb, so, si, True)
    ref_f = ref_b = None
    for how in ('as built', 'largest first', 'smallest first', 'shuffled'):
        sf, sb = reordered(gidx.by_dst.seg, how), reordered(gidx.by_src.seg, how)
        out_f = ops.bdd_aggregate(sf, gidx.nbr_by_dst, ridx.et_by_dst, norm, gidx.by_dst.perm, x, wp, nb, si, so, False, None, 0, packed=True)
        out_b = ops.bdd_aggregate(sb, gidx.nbr_by_src, ridx.et_by_src, norm, gidx.by_src.perm, gg, wpt, nb, so, si, True, packed=True)
        if ref_f is None:
            ref_f, ref_b = out_f, out_b
        same = torch.equal(ref_f, out_f) and torch.equal(ref_b, out_b)
        tf = timeit(lambda: ops.bdd_aggregate(sf, gidx.nbr_by_dst, ridx.et_by_dst, norm, gidx.by_dst.perm, x, wp, nb, si, so, False, None, 0, packed=True))
        tb = timeit(lambda: ops.bdd_aggregate(sb, gidx.nbr_by_src, ridx.et_by_src, norm, gidx.by_src.perm, gg, wpt, nb, so, si, True, packed=True))
        print(f'{si}x{so} forward {tf:7.1f} us, {so}x{si} backward-x {tb:7.1f} us   items {how:14s} (same result: {same})', flush=True)
