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


def _full_step_parity(config, edges, triplets):
    """bench.py's parity leg at the configuration's FULL size: one HIP forward + loss + backward against the oracle's step on
    the same weights and random draws (loss, z, six parameter gradients); raises beyond the configuration's tolerance."""
    from gcn_vae_amd import ops
    bench, args = _bench_args('--config', config)
    dev = torch.device('cuda', 0)
    old = ops.GEMM_PRECISION
    ops.set_gemm_precision(args.gemm_precision)
    try:
        w = bench.make_workload(0, 1, args, dev)
        assert int(w['src'].numel()) == edges and int(w['samples'].shape[0]) == triplets
        model = bench.build_model(w, args).to(dev).train()
        model.static_batch = True
        from gcn_vae_amd.optim import FlatAdam
        opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0)
        inputs = dict(g=w['g'], node_id=w['node_id'].to(dev), etype=w['rel'].to(dev), enorm=w['enorm'],
                      samples=w['samples'].to(dev), labels=w['labels'].to(dev))
        k1_bf = bench.k1_bf16_mode(w, args, dev)
        _, ref = bench.cpu_baseline(w, model, args, budget_s=1.0, k1_bf16=k1_bf)      # 1 warm-up + 2 oracle steps
        rec = bench.parity_check(model, opt, inputs, ref, dev, args.gemm_precision == 'bf16')
        return rec, k1_bf, args
    finally:
        ops.set_gemm_precision(old)


def test_c3_full_training_step_against_oracle():
    """BASELINE configs[2] at FULL WN18RR size (40 943 entities, 22 directed relation types, 173 670 edges, num_bases = 20,
    3 IAF blocks, bf16 operands): the LDS-resident K1 kernel with bf16 operands in both layers and both directions, the
    bf16 MADE chain, the loss head -- against the oracle under oracle.bf16.enabled(k1=True)."""
    rec, k1_bf, args = _full_step_parity('c3', 173670, 220000)
    assert k1_bf and args.n_flows == 3 and args.gemm_precision == 'bf16'
    assert rec['passed'] and rec['outputs_max_rel_err'] <= 5e-3 and rec['gradients_max_rel_l2_err'] <= 2e-2


def test_c4_full_training_step_h500_against_oracle():
    """BASELINE configs[3]'s width (emb_dim = 500: 5x5 / 5x10 blocks on the relation-phase kernel) as a WHOLE training step at
    full FB15k-237 size; the oracle evaluates its per-edge weight gather over edge chunks (oracle.rgcn, 5.4 / 10.9 GB otherwise)."""
    rec, k1_bf, args = _full_step_parity('c4', 544230, 220000)
    assert not k1_bf and args.hidden == 500
    assert rec['passed'] and rec['outputs_max_rel_err'] <= 1e-4 and rec['gradients_max_rel_l2_err'] <= 5e-4


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
    monkeypatch.setattr(ops.indices, 'K1_PHASES', phases)
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


@pytest.mark.parametrize('tag', ['u', 'n'])
def test_device_batch_stages_equal_the_reference_captured_pipeline(tag):
    """SURVEY 8(f-1) against tests/golden/pipeline.npz DIRECTLY (arrays captured from the reference's utils.py under a fixed
    numpy stream): the batch's chosen triplets -- read back out of the golden samples -- go through the device relabel
    (np.unique semantics), the negative sampler fed the draws the golden negatives imply, and the graph builder on the kept
    half; every device array must equal the reference's: uniq_v, relabelled triplets, samples, labels, src, dst, rel, the
    edge norm.  (test_gpu_ops compares the same kernels with the product's host restatement on bigger inputs.)"""
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'pipeline.npz'))
    num_nodes, n_rel, neg, k = 300, 12, 4, 200
    uniq_v, samples, labels = g[f'{tag}_uniq_v'], g[f'{tag}_samples'], g[f'{tag}_labels']
    pos = samples[:k]                                            # relabelled positives, in the sampler's order
    s_orig, o_orig, rel = uniq_v[pos[:, 0]], uniq_v[pos[:, 2]], pos[:, 1]

    def i32(a):
        return torch.from_numpy(np.ascontiguousarray(a).astype(np.int32)).cuda()

    st = lib.stream()
    cap = min(2 * k, num_nodes)
    uniq, src, dst, count = (torch.empty(cap, dtype=torch.int32, device='cuda'), torch.empty(k, dtype=torch.int32, device='cuda'),
                             torch.empty(k, dtype=torch.int32, device='cuda'), torch.empty(1, dtype=torch.int32, device='cuda'))
    wb = int(lib.load().gv_relabel_workspace_bytes(num_nodes))
    ws = torch.empty(wb, dtype=torch.uint8, device='cuda')
    a_g, b_g, rel_d = i32(s_orig), i32(o_orig), i32(rel)
    lib.call('gv_relabel_pairs', ptr(a_g), ptr(b_g), k, num_nodes, ptr(uniq), cap, ptr(src), ptr(dst), ptr(count), ptr(ws), wb, st)
    n = int(count.item())
    assert n == len(uniq_v) and np.array_equal(uniq[:n].cpu().numpy(), uniq_v)
    assert np.array_equal(src.cpu().numpy(), pos[:, 0]) and np.array_equal(dst.cpu().numpy(), pos[:, 2])
    # negatives: the draws the golden rows imply (a row equal to its positive: the draw hit the value it replaced)
    negs, tiled = samples[k:], np.tile(pos, (neg, 1))
    subj = negs[:, 0] != tiled[:, 0]
    obj = negs[:, 2] != tiled[:, 2]
    assert not (subj & obj).any() and np.array_equal(negs[:, 1], tiled[:, 1])
    hit = subj | ~obj
    values = np.where(hit, negs[:, 0], negs[:, 2])
    out_s = torch.empty(k * (neg + 1), 3, dtype=torch.int64, device='cuda')
    out_l = torch.empty(k * (neg + 1), dtype=torch.float32, device='cuda')
    hit_d, val_d = torch.from_numpy(hit.astype(np.uint8)).cuda(), i32(values)
    lib.call('gv_negative_sampling', ptr(src), ptr(rel_d), ptr(dst), k, neg, None, ptr(val_d), ptr(hit_d), 0, 0, None, 0,
             ptr(out_s), ptr(out_l), st)
    assert np.array_equal(out_s.cpu().numpy(), samples) and np.array_equal(out_l.cpu().numpy(), labels)
    # the graph of the kept half: its forward edges (relation < n_rel) name the kept triplets
    gs, gd, gr = g[f'{tag}_src'], g[f'{tag}_dst'], g[f'{tag}_rel']
    fwd = gr < n_rel
    pool = {}
    for i, t in enumerate(map(tuple, pos)):
        pool.setdefault(t, []).append(i)
    keep = np.array([pool[(int(a), int(r_), int(b))].pop() for a, r_, b in zip(gs[fwd], gr[fwd], gd[fwd])])
    m = len(keep)
    assert m == k // 2 and len(gs) == 2 * m
    src2, dst2, rel2 = (torch.empty(2 * m, dtype=torch.int32, device='cuda') for _ in range(3))
    norm = torch.empty(2 * m, dtype=torch.float32, device='cuda')
    gb = int(lib.load().gv_graph_from_triplets_workspace_bytes(m, cap, n_rel))
    gws = torch.empty(gb, dtype=torch.uint8, device='cuda')
    keep_d = i32(keep)
    lib.call('gv_graph_from_triplets', ptr(src), ptr(rel_d), ptr(dst), ptr(keep_d), m, cap, n_rel, ptr(src2), ptr(dst2), ptr(rel2),
             ptr(norm), ptr(gws), gb, st)
    assert np.array_equal(src2.cpu().numpy(), gs) and np.array_equal(dst2.cpu().numpy(), gd) and np.array_equal(rel2.cpu().numpy(), gr)
    assert np.array_equal(norm.cpu().numpy(), g[f'{tag}_edge_norm'].reshape(-1))       # 1 / in-degree of the destination, bit for bit
    assert np.array_equal(g[f'{tag}_norm'][gd], g[f'{tag}_edge_norm'].reshape(-1))


@pytest.mark.parametrize('n_flows', [0, 2])
def test_kgvae_sample_z_against_oracle_with_the_same_draws(n_flows):
    """KGVAE.sample_z (kgvae/model.py:60-69: mixture component by Categorical(pi), reparameterised draw, the flows' inverse
    chain) on the HIP modules against oracle.kgvae.sample_z fed the very draws the device generator produced."""
    from gcn_vae_amd.encoders import KGVAE
    from oracle import kgvae as okg
    torch.manual_seed(0)
    enc = KGVAE(50, 16, 16, 6, num_bases=4, num_hidden_layers=2, dropout=0.0, use_self_loop=True, use_cuda=True, k=5,
                n_flows=n_flows).cuda()
    with torch.no_grad():
        enc.z_pre.normal_(0, 0.7)
        for p in enc.nf.parameters() if n_flows else []:
            p.mul_(1.5)
    batch = 37
    torch.manual_seed(123)
    with torch.no_grad():
        got = enc.sample_z(batch)
    torch.manual_seed(123)                                           # replay the two draws in the order sample_z makes them
    idx = torch.distributions.categorical.Categorical(enc.pi).sample((batch,))
    eps = torch.randn(batch, 16, device='cuda')
    state = {'encoder.' + k: v.detach().cpu() for k, v in enc.state_dict().items()}
    want = okg.sample_z(state, idx.cpu(), eps.cpu(), n_flows)
    assert got.shape == (batch, 16)
    close(got, want, rtol=2e-4, atol_scale=2e-5, msg='sample_z')
