"""BASELINE.json configs at their OWN sizes on the GPU (pytest -m gpu): configs[1] as a whole training step, configs[3]'s
layers at FB15k-237 size with emb_dim = 500, configs[4]'s 1 M-entity / 50 M-edge graph through size-independent
properties plus oracle equality on sampled rows.  The oracle (oracle/) is the checker; tolerances as in test_gpu_ops."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def close(a, b, rtol=1e-4, atol_scale=1e-5, msg=''):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    atol = atol_scale * max(1.0, float(b.abs().max()) if b.numel() else 1.0)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, msg=lambda m: f'{msg}: {m}')


def _bench_args(*argv):
    import bench
    old = sys.argv
    sys.argv = ['bench.py'] + list(argv)
    try:
        return bench, bench.parse()
    finally:
        sys.argv = old


def test_c2_full_training_step_against_oracle():
    """The timed workload of bench.py (BASELINE configs[1]: FB15k-237-shaped full graph, both R-GCN layers over 544 230
    edges, reparameterisation, KL + MMD, DistMult + BCE on T = 220 000 triplets) as ONE forward + loss + backward on the
    HIP path against the oracle's step on the same weights and random draws: loss, z and six parameter gradients."""
    bench, args = _bench_args('--config', 'c2')
    dev = torch.device('cuda', 0)
    w = bench.make_workload(0, 1, args, dev)
    assert int(w['src'].numel()) == 544230 and int(w['samples'].shape[0]) == 220000
    model = bench.build_model(w, args).to(dev).train()
    model.static_batch = True
    from gcn_vae_amd.optim import FlatAdam
    opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0)
    inputs = dict(g=w['g'], node_id=w['node_id'].to(dev), etype=w['rel'].to(dev), enorm=w['enorm'],
                  samples=w['samples'].to(dev), labels=w['labels'].to(dev))
    _, ref = bench.cpu_baseline(w, model, args, budget_s=1.0)          # 1 warm-up + 2 oracle steps
    rec = bench.parity_check(model, opt, inputs, ref, dev)             # raises beyond 1e-4 / 5e-4
    assert rec['passed'] and rec['outputs_max_rel_err'] <= 1e-4 and rec['gradients_max_rel_l2_err'] <= 5e-4
    assert len([k for k in rec['checked'] if k.startswith('grad ')]) >= 3


def _oracle_layer_chunked(x, src, dst, et, norm, p, nb, gout, chunk=40000):
    """oracle.rgcn.rel_graph_conv (bdd, identity activation) without its E x (in*out/B) weight gather alive at once:
    the aggregate is linear in the messages, so it is summed over edge chunks (forward, no graph), the non-linear tail
    runs once, and each chunk is then back-propagated on its own (leaf gradients accumulate)."""
    from oracle import rgcn as orgcn
    n = x.shape[0]
    out_feat = p['loop_weight'].shape[1]
    agg = torch.zeros(n, out_feat)
    with torch.no_grad():
        for c0 in range(0, src.numel(), chunk):
            sl = slice(c0, c0 + chunk)
            agg.index_add_(0, dst[sl], orgcn._messages(x, src[sl], et[sl], norm[sl], p, 'bdd', nb))
    xo = x.clone().requires_grad_(True)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    aggo = agg.clone().requires_grad_(True)
    h = aggo + po['h_bias'] + xo @ po['loop_weight']
    h.backward(gout)
    g_agg = aggo.grad
    for c0 in range(0, src.numel(), chunk):
        sl = slice(c0, c0 + chunk)
        msg = orgcn._messages(xo, src[sl], et[sl], norm[sl], po, 'bdd', nb)
        msg.backward(g_agg.index_select(0, dst[sl]))
    return h.detach(), xo.grad, {k: v.grad for k, v in po.items()}


@pytest.mark.parametrize('fin,fout', [(500, 500), (500, 1000)])
def test_c4_layer_full_size_h500_against_oracle(fin, fout):
    """BASELINE configs[3] / the reference's default --n-hidden 500 at FULL FB15k-237 size: one R-GCN layer (B = 100:
    5x5 or 5x10 blocks, 474 relation types, 544 230 edges), forward and every gradient, against the oracle (evaluated
    over edge chunks).  All four aggregation launches run on the relation-phase kernel (the per-row kernels are covered by test_gpu_ops)."""
    from gcn_vae_amd import ops, sampling
    from gcn_vae_amd.data import FB15K237, synthetic_kg
    from oracle import rgcn as orgcn
    data = synthetic_kg(FB15K237['num_nodes'], FB15K237['num_rels'], FB15K237['n_train'], seed=0)
    graph, rel, node_norm = sampling.build_test_graph(data.num_nodes, data.num_rels, data.train)
    src, dst = graph.edges()
    n, r, nb = data.num_nodes, 2 * data.num_rels, 100
    norm = torch.from_numpy(node_norm)[dst].view(-1, 1)
    gen = torch.Generator().manual_seed(fin + fout)
    x = torch.randn(n, fin, generator=gen)
    p = orgcn.init_params(fin, fout, r, 'bdd', nb, True, True, gen)
    p['h_bias'] = torch.randn(fout, generator=gen) * 0.1
    gout = torch.randn(n, fout, generator=gen)
    et = torch.from_numpy(rel)
    ho, gxo, gpo = _oracle_layer_chunked(x, src, dst, et, norm, p, nb, gout)
    gidx = graph.device_index('cuda')
    ridx = gidx.relation_index(et.cuda(), r)
    if fout == 1000:
        assert ops.use_phases(gidx, 5, 10, False, n, fin) and ops.use_phases(gidx, 10, 5, True, n, fout)
    else:
        assert ops.use_phases(gidx, 5, 5, False, n, fin) and ops.use_phases(gidx, 5, 5, True, n, fout)
    xg = x.cuda().requires_grad_(True)
    pg = {k: v.cuda().requires_grad_(True) for k, v in p.items()}
    hg = ops.rel_graph_conv_bdd(xg, pg['weight'], pg['h_bias'], pg['loop_weight'], norm.cuda(), gidx, ridx, nb, 0)
    hg.backward(gout.cuda())
    close(hg, ho, msg='forward')
    close(xg.grad, gxo, msg='grad_x')
    close(pg['weight'].grad, gpo['weight'], msg='grad_weight')
    close(pg['loop_weight'].grad, gpo['loop_weight'], rtol=3e-4, atol_scale=3e-5, msg='grad_loop')
    close(pg['h_bias'].grad, gpo['h_bias'], rtol=3e-4, atol_scale=3e-5, msg='grad_bias')


@pytest.mark.parametrize('phases,so', [('0', 2), ('auto', 2), ('auto', 4)])
def test_c5_scale_properties_and_sampled_rows(monkeypatch, phases, so):
    """BASELINE configs[4] on ONE GPU: 1 M entities, 50 M directed edges, 2 000 relation types, emb_dim 200, B = 100
    (9-10 GiB).  The oracle cannot materialise this graph, so: (1) linearity of the aggregate in x, (2) invariance to the
    order the edges are handed in, (3) the backward-x aggregation as the adjoint of the forward one, (4) equality with
    the oracle on 256 sampled destination rows (the oracle on those rows' in-edges only).  phases = '0': per-row
    kernels, 'auto': the relation-phase kernel (the 800 MB feature table is HBM scale); so = 4 runs layer 2's block shapes."""
    from gcn_vae_amd import ops
    from oracle import rgcn as orgcn
    monkeypatch.setattr(ops, 'K1_PHASES', phases)
    n, e, r, nb, si = 1_000_000, 50_000_000, 2000, 100, 2          # so = 4: layer 2's 2x4 blocks (forward) / 4x2 (adjoint)
    dev = torch.device('cuda', 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    src = (torch.rand(e, device=dev, generator=gen) ** 2 * n).long().clamp_(max=n - 1)
    dst = (torch.rand(e, device=dev, generator=gen) ** 2 * n).long().clamp_(max=n - 1)
    et = torch.randint(0, r, (e,), device=dev, generator=gen)
    deg = torch.bincount(dst, minlength=n).float()
    norm = (1.0 / deg.clamp(min=1))[dst].contiguous()
    w = torch.randn(r, nb * si * so, device=dev, generator=gen) * 0.3
    x1 = torch.randn(n, nb * si, device=dev, generator=gen)
    x2 = torch.randn(n, nb * si, device=dev, generator=gen)
    gidx = ops.GraphIndex(src, dst, n)
    ridx = gidx.relation_index(et, r)
    use_ph = ops.use_phases(gidx, si, so, False, n, nb * si)
    assert use_ph == (phases == 'auto')

    def agg(g_, r_, xx, coef, trans=False):
        side = 'src' if trans else 'dst'
        p, q = (so, si) if trans else (si, so)          # gathered / produced block width of this launch
        if use_ph:
            ph = r_.phase_order(g_, side, nb, p, q)
            return ops.bdd_aggregate_phases(ph, ph.coef(coef), xx, ops.pack_weight_phase(ph, w, nb, p, q), r, nb, p, q)
        order = g_.by_src if trans else g_.by_dst
        return ops.bdd_aggregate(order.seg, g_.nbr_by_src if trans else g_.nbr_by_dst, r_.et_by_src if trans else r_.et_by_dst,
                                 coef, order.perm, xx, w, nb, p, q, trans)

    a1, a2 = agg(gidx, ridx, x1, norm), agg(gidx, ridx, x2, norm)
    scale = float(a1.abs().max())
    # (1) linearity
    a12 = agg(gidx, ridx, 2.0 * x1 - 0.5 * x2, norm)
    assert float((a12 - (2.0 * a1 - 0.5 * a2)).abs().max()) < 1e-4 * max(1.0, scale)
    del a12, a2
    # (3) adjoint: <agg(x1), y> == <x1, agg^T(y)>
    y = torch.randn(n, nb * so, device=dev, generator=gen)
    at = agg(gidx, ridx, y, norm, trans=True)
    lhs, rhs = float((a1.double() * y.double()).sum()), float((x1.double() * at.double()).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    del at, y
    # (4) sampled destination rows against the oracle on their in-edges
    rows = torch.randint(0, n, (256,), device=dev, generator=gen).unique()
    rp = gidx.by_dst.seg.rowptr.long()
    pos = torch.cat([torch.arange(int(rp[v]), int(rp[v + 1]), device=dev) for v in rows.tolist()])
    eid = pos if gidx.by_dst.perm is None else gidx.by_dst.perm.long()[pos]
    local = torch.searchsorted(rows, dst[eid])
    assert bool((rows[local] == dst[eid]).all())
    src_c = src[eid].cpu()
    uniq, inv = torch.unique(src_c, return_inverse=True)
    x_sub = torch.cat([x1.cpu()[uniq], torch.zeros(max(0, rows.numel() - uniq.numel()), nb * si)])
    msg = orgcn._messages(x_sub, inv, et[eid].cpu(), norm[eid].cpu(), {'weight': w.cpu()}, 'bdd', nb)
    want = torch.zeros(rows.numel(), nb * so).index_add(0, local.cpu(), msg)
    close(a1[rows], want, msg='sampled rows')
    # (2) edge-order invariance: the same multiset of edges handed over in a shuffled order
    perm = torch.randperm(e, device=dev, generator=gen)
    g2 = ops.GraphIndex(src[perm], dst[perm], n)
    r2 = g2.relation_index(et[perm].contiguous(), r)
    b1 = agg(g2, r2, x1, norm[perm].contiguous())
    assert float((b1 - a1).abs().max()) < 1e-4 * max(1.0, scale)
