"""Whole-path parity on the GPU: modules with the reference's signatures, loaded with the reference's
state_dict, against (a) the golden vectors captured from the reference at C1 size and (b) the CPU
oracle at the C2 mini-batch shape.   pytest -m gpu.   Tolerance 1e-4 (north_star)."""
import random

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import kgvae as okg

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-4, atol_scale=1e-5, msg=''):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    atol = atol_scale * max(1.0, float(b.abs().max()) if b.numel() else 1.0)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, msg=lambda m: f'{msg}: {m}')


def build_model(g, n_flows, kl, mmd, h=16, nb=4, num_nodes=1000, num_rels=20, dropout=0.0):
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.train import LinkPredict
    net = LinkPredict(KGVAE, num_nodes, h, num_rels, num_bases=nb, num_hidden_layers=2, dropout=dropout,
                      use_cuda=True, reg_param=0.01, kl_param=kl, mmd_param=mmd, k=10, n_flows=n_flows)
    state = {k[6:]: v for k, v in g.items() if k.startswith('state.')}
    net.load_state_dict(state)           # strict: proves key/shape parity with the reference
    return net.cuda()


def run_golden(tag, n_flows, kl, mmd):
    from gcn_vae_amd.graph import KGraph
    g = load_golden(f'model_c1_{tag}.npz')
    net = build_model(g, n_flows, kl, mmd)
    net.train()
    graph = KGraph()
    graph.add_nodes(len(g['node_id']))
    graph.add_edges(g['src'], g['dst'])
    enc = net.encoder
    enc.eps_override = g['eps'].cuda()
    enc.mmd_eps_override = g['eps_prior'].cuda()
    enc.mmd_index_override = g['post_idx'].cuda()
    embed = net(graph, g['node_id'].view(-1, 1).cuda(), g['etype'].cuda(), g['edge_norm'].cuda())
    loss, pred, klv, mmdv = net.get_loss(graph, embed, g['samples'].cuda(), g['labels'].cuda())
    loss.backward()
    return g, net, embed, (loss, pred, klv, mmdv)


def test_golden_c1_with_flows():
    g, net, embed, (loss, pred, kl, mmd) = run_golden('flows3', 3, 1e-5, 1.0)
    close(embed, g['z'], msg='z')
    close(net.encoder.z_mean, g['z_mean'], msg='z_mean')
    close(net.encoder.z_sigma, g['z_sigma'], msg='z_sigma')
    close(net.encoder.get_flow_log_prob(), g['flow_log_prob'], msg='flow_log_prob')
    for a, b in ((loss, 'loss'), (pred, 'pred'), (kl, 'kl'), (mmd, 'mmd')):
        close(a, g[b], msg=b)
    close(net.calc_score(embed, g['samples'].cuda()), g['score'], msg='score')
    close(net.regularization_loss(embed), g['reg'], msg='reg')
    n = 0
    for name, p in net.named_parameters():
        if 'grad.' + name in g:
            close(p.grad, g['grad.' + name], rtol=2e-4, atol_scale=2e-5, msg='grad ' + name)
            n += 1
    assert n >= 36


def test_golden_c1_without_flows():
    g, net, embed, (loss, pred, kl, mmd) = run_golden('flows0', 0, 0.0, 1.0)
    close(embed, g['z'], msg='z')
    for a, b in ((loss, 'loss'), (pred, 'pred'), (mmd, 'mmd')):
        close(a, g[b], msg=b)
    for name, p in net.named_parameters():
        if 'grad.' + name in g:
            close(p.grad, g['grad.' + name], rtol=2e-4, atol_scale=2e-5, msg='grad ' + name)
        elif p.requires_grad:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name


def test_made_module_against_golden():
    import seeded
    from gcn_vae_amd.flows import MADE, PermuteLayer
    g = load_golden('made.npz')
    for tag, (d, h, nh, seed) in {'d16': (16, 16, 3, 100), 'd200': (200, 200, 3, 200), 'd8h12': (8, 12, 2, 300)}.items():
        m = MADE(d, h, nh)
        with torch.no_grad():
            for i, (w, b) in enumerate(seeded.made_layers(seed, d, h, nh)):
                m.net[2 * i].weight.copy_(w)
                m.net[2 * i].bias.copy_(b)
        m = m.cuda()
        z = g[f'{tag}_z'].cuda().requires_grad_(True)
        x, ld = m.forward(z)
        close(x, g[f'{tag}_x'], msg=tag + ' x')
        close(ld, g[f'{tag}_logdet'], msg=tag + ' logdet')
        (x.pow(2).sum() + ld.sum()).backward()
        close(z.grad, g[f'{tag}_grad_z'], rtol=2e-4, atol_scale=2e-5, msg=tag + ' grad_z')
        close(m.net[0].weight.grad, g[f'{tag}_grad_w0'], rtol=2e-4, atol_scale=2e-5, msg=tag + ' grad_w0')
        close(m.net[2 * (nh + 1)].weight.grad, g[f'{tag}_grad_wlast'], rtol=2e-4, atol_scale=2e-5, msg=tag + ' grad_wlast')
        zi, ldi = m.inverse(g[f'{tag}_z'].cuda())
        close(zi, g[f'{tag}_inv'], msg=tag + ' inverse')
        close(ldi, g[f'{tag}_inv_logdet'], msg=tag + ' inverse logdet')
    px, pld = PermuteLayer(7)(g['perm_in'].cuda())
    close(px, g['perm_out'])
    assert pld.shape == (3, 1) and float(pld.abs().max()) == 0


@pytest.mark.parametrize('n_flows', [0, 3])
def test_c2_minibatch_against_oracle(n_flows):
    """FB15k-237-shaped mini-batch (h=200, B=100, 474 directed relations, E=20k, T=44k) incl. dropout masks."""
    from gcn_vae_amd import sampling
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.train import LinkPredict
    data = synthetic_kg(14541, 237, 60000, seed=1)
    adj, deg = sampling.get_adj_and_degrees(data.num_nodes, data.train)
    np.random.seed(0)
    graph, node_id, etype, node_norm, samples, labels = sampling.generate_sampled_graph_and_labels(
        data.train, 20000, 0.5, data.num_rels, adj, deg, 1 if n_flows else 3, 'uniform')
    n = len(node_id)
    torch.manual_seed(0)
    kl_param = 1e-2
    net = LinkPredict(KGVAE, data.num_nodes, 200, data.num_rels, num_bases=100, num_hidden_layers=2, dropout=0.2,
                      use_cuda=True, reg_param=0.01, kl_param=kl_param, mmd_param=1.0, k=10, n_flows=n_flows)
    with torch.no_grad():
        net.encoder.rconv_layer_1.h_bias.normal_(0, 0.1)
        net.encoder.rconv_layer_2.h_bias.normal_(0, 0.1)
    state = {k: v.detach().clone().requires_grad_(v.is_floating_point() and 'mask' not in k and not k.endswith('.pi'))
             for k, v in net.state_dict().items()}
    gen = torch.Generator().manual_seed(5)
    eps, eps_prior = torch.randn(n, 200, generator=gen), torch.randn(200, 200, generator=gen)
    keep1 = (torch.rand(n, 200, generator=gen) > 0.2).to(torch.uint8)
    keep2 = (torch.rand(n, 400, generator=gen) > 0.2).to(torch.uint8)
    random.seed(3)
    post_idx = torch.tensor(random.sample(range(n), 200))
    src, dst = graph.edges()
    node_id_t = torch.from_numpy(node_id).view(-1, 1)
    etype_t = torch.from_numpy(etype)
    enorm = sampling.node_norm_to_edge_norm(graph, torch.from_numpy(node_norm).view(-1, 1))
    samples_t, labels_t = torch.from_numpy(samples), torch.from_numpy(labels)
    enc = okg.kgvae_encode(state, src, dst, node_id_t, etype_t, enorm, eps, 100, n_flows, 0.2, keep1, keep2)
    lo = okg.link_predict_loss(state, enc, samples_t, labels_t, 0.01, kl_param, 1.0, 10, n_flows, eps_prior, post_idx)
    lo[0].backward()

    net = net.cuda().train()
    e = net.encoder
    e.eps_override, e.mmd_eps_override, e.mmd_index_override = eps.cuda(), eps_prior.cuda(), post_idx.cuda()
    e.rconv_layer_1.keep_mask_override, e.rconv_layer_2.keep_mask_override = keep1.cuda(), keep2.cuda()
    embed = net(graph, node_id_t.cuda(), etype_t.cuda(), enorm.cuda())
    lg = net.get_loss(graph, embed, samples_t.cuda(), labels_t.cuda())
    lg[0].backward()
    close(embed, enc['z'], msg='z')
    for a, b, nm in zip(lg, lo, ('loss', 'pred', 'kl', 'mmd')):
        close(a, b, msg=nm)
    for name, p in net.named_parameters():
        ref = state[name].grad
        if ref is None:
            continue
        close(p.grad, ref, rtol=5e-4, atol_scale=5e-5, msg='grad ' + name)


def test_rgcn_encoder_and_eval_ranking():
    """--model-class RGCN (crashes in the reference, SURVEY 0 bug 2) + the GEMM-based ranker vs the oracle's."""
    from gcn_vae_amd import ranking, sampling
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.encoders import RGCN
    from gcn_vae_amd.train import LinkPredict
    from oracle import ranking as orank
    data = synthetic_kg(300, 7, 1500, 120, seed=2)
    torch.manual_seed(1)
    net = LinkPredict(RGCN, data.num_nodes, 16, data.num_rels, num_bases=4, num_hidden_layers=2, dropout=0.0,
                      use_cuda=True, reg_param=0.01)
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    graph, rel, norm = sampling.build_test_graph(data.num_nodes, data.num_rels, data.train)
    src, dst = graph.edges()
    node_id = torch.arange(data.num_nodes).view(-1, 1)
    enorm = sampling.node_norm_to_edge_norm(graph, torch.from_numpy(norm).view(-1, 1))
    ho = okg.rgcn_encode(state, src, dst, node_id, torch.from_numpy(rel), enorm, 4, 2)
    net = net.cuda().eval()
    hg = net(graph, node_id.cuda(), torch.from_numpy(rel).cuda(), enorm.cuda())
    close(hg, ho, msg='rgcn embed')
    valid = torch.from_numpy(data.valid)
    mrr_o, _, ranks_o = orank.calc_mrr(ho, state['w_relation'], valid, hits=[1, 3, 10], eval_bz=50,
                                       flow_log_prob=torch.tensor(0.0))
    mrr_g = ranking.calc_mrr(hg, net.w_relation, valid.cuda(), hits=[1, 3, 10], eval_bz=50, verbose=False)
    assert abs(mrr_o - mrr_g) < 2e-3, (mrr_o, mrr_g)


def test_separate_loss_ops_match_fused_head():
    """calc_score / regularization_loss / get_kl / get_mmd (reference signatures) agree with get_loss's fused node."""
    g, net, embed, (loss, pred, kl, mmd) = run_golden('flows3', 3, 1e-5, 1.0)
    enc = net.encoder
    embed2 = embed.detach()
    score = net.calc_score(embed2, g['samples'].cuda()) + enc.get_flow_log_prob().detach()
    pred2 = torch.nn.functional.binary_cross_entropy_with_logits(score, g['labels'].cuda())
    close(pred2, pred, msg='pred')
    close(net.regularization_loss(embed2), g['reg'], msg='reg')
    close(enc.get_kl(embed2), kl, msg='kl')
    close(enc.get_mmd(embed2), mmd, atol_scale=1e-6, msg='mmd')


def test_c2_full_graph_layer_against_oracle():
    """BASELINE configs[1] at FULL size: one R-GCN layer (h=200 -> 400, B=100, 474 relation types) over the
    544 230-edge FB15k-237-shaped graph, forward and all gradients, against the CPU oracle."""
    from gcn_vae_amd import ops, sampling
    from gcn_vae_amd.data import FB15K237, synthetic_kg
    from oracle import rgcn as orgcn
    data = synthetic_kg(FB15K237['num_nodes'], FB15K237['num_rels'], FB15K237['n_train'], seed=0)
    graph, rel, node_norm = sampling.build_test_graph(data.num_nodes, data.num_rels, data.train)
    src, dst = graph.edges()
    n, r = data.num_nodes, 2 * data.num_rels
    assert src.numel() == 544230
    norm = torch.from_numpy(node_norm)[dst].view(-1, 1)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(n, 200, generator=gen)
    p = orgcn.init_params(200, 400, r, 'bdd', 100, True, True, gen)
    p['h_bias'] = torch.randn(400, generator=gen) * 0.1
    gout = torch.randn(n, 400, generator=gen)
    et = torch.from_numpy(rel)
    xo = x.clone().requires_grad_(True)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    orgcn.rel_graph_conv(xo, src, dst, et, norm, po, 'bdd', 100, torch.relu).backward(gout)
    ho = orgcn.rel_graph_conv(x, src, dst, et, norm, p, 'bdd', 100, torch.relu)
    gidx = graph.device_index('cuda')
    ridx = gidx.relation_index(et.cuda(), r)
    xg = x.cuda().requires_grad_(True)
    pg = {k: v.cuda().requires_grad_(True) for k, v in p.items()}
    hg = ops.rel_graph_conv_bdd(xg, pg['weight'], pg['h_bias'], pg['loop_weight'], norm.cuda(), gidx, ridx, 100, 1)
    hg.backward(gout.cuda())
    close(hg, ho, msg='forward')
    close(xg.grad, xo.grad, msg='grad_x')
    close(pg['weight'].grad, po['weight'].grad, msg='grad_weight')
    close(pg['loop_weight'].grad, po['loop_weight'].grad, rtol=3e-4, atol_scale=3e-5, msg='grad_loop')
    close(pg['h_bias'].grad, po['h_bias'].grad, rtol=3e-4, atol_scale=3e-5, msg='grad_bias')


def test_large_graph_properties():
    """4 M directed edges over 200 k entities (beyond what the oracle materialises in seconds): size-independent
    properties of the aggregation -- linearity in x, invariance to the edge order handed in, and total mass."""
    from gcn_vae_amd import ops
    n, e, r, nb, si, so = 200_000, 4_000_000, 400, 100, 2, 4
    gen = torch.Generator(device='cuda').manual_seed(0)
    src = torch.randint(0, n, (e,), device='cuda', generator=gen)
    dst = (torch.rand(e, device='cuda', generator=gen) ** 3 * n).long().clamp_(max=n - 1)     # skewed in-degrees
    et = torch.randint(0, r, (e,), device='cuda', generator=gen)
    deg = torch.bincount(dst, minlength=n).float()
    norm = (1.0 / deg.clamp(min=1))[dst]
    w = torch.randn(r, nb * si * so, device='cuda', generator=gen)
    x1 = torch.randn(n, nb * si, device='cuda', generator=gen)
    x2 = torch.randn(n, nb * si, device='cuda', generator=gen)
    gidx = ops.GraphIndex(src, dst, n)
    assert gidx.by_dst.perm is not None and gidx.by_dst.seg.n_fix > 0          # unsorted input, hub rows split
    ridx = gidx.relation_index(et, r)

    def agg(xx, gi=gidx, ri=ridx, nm=norm):
        return ops.bdd_aggregate(gi.by_dst.seg, gi.nbr_by_dst, ri.et_by_dst, nm, gi.by_dst.perm, xx, w, nb, si, so)
    a1, a2, a12 = agg(x1), agg(x2), agg(2.0 * x1 - 0.5 * x2)
    close(a12, 2.0 * a1 - 0.5 * a2, rtol=1e-4, atol_scale=2e-5, msg='linearity')
    perm = torch.randperm(e, device='cuda', generator=gen)
    g2 = ops.GraphIndex(src[perm], dst[perm], n)
    r2 = g2.relation_index(et[perm], r)
    close(agg(x1, g2, r2, norm[perm]), a1, rtol=1e-4, atol_scale=2e-5, msg='edge-order invariance')
    # total mass: sum_v out[v] = sum_e norm_e * blockdiag(W_e) x[src_e], evaluated edge-wise in fp64 on a sample of columns
    cols = torch.tensor([0, 1, 7, 202, 399], device='cuda')
    b, j = cols // so, cols % so
    wsel = w.view(r, nb, si, so).double()
    msg = torch.zeros(e, len(cols), dtype=torch.float64, device='cuda')
    for i in range(si):
        msg += x1[src][:, b * si + i].double() * wsel[et][:, b, i, j]
    tot = (msg * norm.double().unsqueeze(1)).sum(0)
    close(a1[:, cols].double().sum(0), tot, rtol=1e-4, atol_scale=1e-5, msg='total mass')


def test_made_fused_node_equals_op_chain():
    from gcn_vae_amd.flows import MADE
    torch.manual_seed(3)
    m = MADE(40, 56, 2).cuda()
    z = torch.randn(300, 40, device='cuda')
    outs = []
    for fn in (m.forward, m.forward_unfused):
        m.zero_grad()
        zz = z.clone().requires_grad_(True)
        x, ld = fn(zz)
        (x.sin().sum() + (ld * ld).sum()).backward()
        outs.append((x, ld, zz.grad, [p.grad.clone() for p in m.parameters()]))
    close(outs[0][0], outs[1][0], msg='x')
    close(outs[0][1], outs[1][1], msg='logdet')
    close(outs[0][2], outs[1][2], rtol=2e-4, atol_scale=2e-5, msg='grad z')
    for a, b in zip(outs[0][3], outs[1][3]):
        close(a, b, rtol=2e-4, atol_scale=2e-5, msg='grad param')


def test_device_sampler_batches_are_well_formed_and_train():
    """device-side batch preparation.  Not the reference's RNG stream, so the checks are structural:
    the batch is what generate_sampled_graph_and_labels would build from the same picks."""
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.device_sampling import DeviceSampler
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.optim import FlatAdam
    from gcn_vae_amd.train import LinkPredict
    data = synthetic_kg(2000, 15, 30000, seed=4)
    sm = DeviceSampler(data.train, data.num_nodes, data.num_rels, 'cuda', seed=1)
    b = sm.sample(4000, 0.5, 3)
    n = len(b.g)
    src, dst = (t.cuda() for t in b.g.edges())
    assert b.node_id.shape == (n, 1) and bool((b.node_id[1:] > b.node_id[:-1]).all())           # unique, sorted global ids
    assert src.numel() == 4000 and b.edge_type.numel() == 4000 and b.edge_norm.shape == (4000, 1)     # 2 x split edges
    key = (dst * n + src) * 30 + b.edge_type
    assert bool((key[1:] >= key[:-1]).all())                                                      # (dst, src, rel) order
    deg = torch.bincount(dst, minlength=n).float()
    assert torch.equal(b.edge_norm.view(-1), (1.0 / deg)[dst])
    # forward edges (rel < R) have a mirrored reverse edge with rel + R
    fwd = b.edge_type < 15
    a = torch.stack([src[fwd], dst[fwd], b.edge_type[fwd]], 1)
    r = torch.stack([dst[~fwd], src[~fwd], b.edge_type[~fwd] - 15], 1)
    assert torch.equal(torch.unique(a, dim=0), torch.unique(r, dim=0))
    # samples: positives are real triplets (after mapping back), negatives differ from their positive in exactly one end
    pos, neg = b.samples[:4000], b.samples[4000:]
    glob = torch.stack([b.node_id.view(-1)[pos[:, 0]], pos[:, 1], b.node_id.view(-1)[pos[:, 2]]], 1)
    train = torch.from_numpy(data.train).cuda()
    keyf = lambda t: (t[:, 0] * 2000 + t[:, 2]) * 15 + t[:, 1]
    assert bool(torch.isin(keyf(glob), keyf(train)).all())
    rep = pos.repeat(3, 1)
    same_s, same_o = neg[:, 0] == rep[:, 0], neg[:, 2] == rep[:, 2]
    assert bool((same_s | same_o).all()) and bool((neg[:, 1] == rep[:, 1]).all())
    assert 0.35 < float((~same_s).float().mean()) < 0.65
    assert b.labels.sum().item() == 4000 and b.labels.numel() == 16000
    # and a few optimisation steps run and reduce the loss
    torch.manual_seed(0)
    net = LinkPredict(KGVAE, data.num_nodes, 16, data.num_rels, num_bases=4, num_hidden_layers=2, dropout=0.1, use_cuda=True,
                      reg_param=0.01, kl_param=1e-5, mmd_param=1.0, k=4, n_flows=1).cuda().train()
    opt = FlatAdam(net.parameters(), lr=1e-2, max_grad_norm=1.0)
    losses = []
    for _ in range(12):
        bb = sm.sample(4000, 0.5, 3)
        opt.zero_grad()
        emb = net(bb.g, bb.node_id, bb.edge_type, bb.edge_norm)
        loss = net.get_loss(bb.g, emb, bb.samples, bb.labels)[0]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    g, node_id, et, en = sm.full_graph()
    assert len(g) == 2000 and g.number_of_edges() == 60000 and node_id.shape == (2000, 1)


def _wn18rr_shaped_case(n_nodes=4000, n_rel=11, n_trip=9000, h=200, nb=20, n_flows=3):
    """BASELINE configs[2] in shape (WN18RR: 11 relations -> 22 directed types, so num_bases = 20 and 10x10 / 10x20
    blocks; 3 IAF blocks), scaled down in node count so that the CPU oracle finishes in seconds."""
    from gcn_vae_amd import sampling
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.train import LinkPredict
    data = synthetic_kg(n_nodes, n_rel, n_trip, seed=3)
    graph, rel, node_norm = sampling.build_test_graph(data.num_nodes, data.num_rels, data.train)
    torch.manual_seed(0)
    net = LinkPredict(KGVAE, data.num_nodes, h, data.num_rels, num_bases=nb, num_hidden_layers=2, dropout=0.0,
                      use_cuda=True, reg_param=0.01, kl_param=1e-3, mmd_param=1.0, k=10, n_flows=n_flows)
    state = {k: v.detach().clone().requires_grad_(v.is_floating_point() and 'mask' not in k and not k.endswith('.pi'))
             for k, v in net.state_dict().items()}
    gen = torch.Generator().manual_seed(9)
    eps, eps_prior = torch.randn(n_nodes, h, generator=gen), torch.randn(200, h, generator=gen)
    random.seed(4)
    post_idx = torch.tensor(random.sample(range(n_nodes), 200))
    src, dst = graph.edges()
    node_id = torch.arange(n_nodes).view(-1, 1)
    etype = torch.from_numpy(rel)
    enorm = sampling.node_norm_to_edge_norm(graph, torch.from_numpy(node_norm).view(-1, 1))
    np.random.seed(1)
    pos = data.train[np.random.choice(len(data.train), 3000, replace=False)]
    samples, labels = sampling.negative_sampling(pos, n_nodes, 2)
    return dict(net=net, state=state, graph=graph, src=src, dst=dst, node_id=node_id, etype=etype, enorm=enorm, eps=eps,
                eps_prior=eps_prior, post_idx=post_idx, samples=torch.from_numpy(samples), labels=torch.from_numpy(labels),
                nb=nb, n_flows=n_flows)


def _oracle_step(c):
    state = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in c['state'].items()}
    enc = okg.kgvae_encode(state, c['src'], c['dst'], c['node_id'], c['etype'], c['enorm'], c['eps'], c['nb'],
                           c['n_flows'], 0.0, None, None)
    lo = okg.link_predict_loss(state, enc, c['samples'], c['labels'], 0.01, 1e-3, 1.0, 10, c['n_flows'], c['eps_prior'],
                               c['post_idx'])
    lo[0].backward()
    return state, enc, lo


def test_c3_wn18rr_shape_bf16_operand_gemms():
    """configs[2]: every dense product (MaskedLinear, self-loop term) with bf16 operands and fp32 accumulation.
    Checked against the oracle's emulation of exactly that (oracle/bf16.py; tolerance 5e-3: a last-bit difference in an
    fp32 activation can flip its bf16 rounding, 2^-9 relative on that operand) and, loosely, against the fp32 oracle."""
    from gcn_vae_amd import ops
    from oracle import bf16
    c = _wn18rr_shaped_case()
    with bf16.enabled():
        st_b, enc_b, lo_b = _oracle_step(c)
    st_f, enc_f, lo_f = _oracle_step(c)
    net = c['net'].cuda().train()
    e = net.encoder
    e.eps_override, e.mmd_eps_override, e.mmd_index_override = c['eps'].cuda(), c['eps_prior'].cuda(), c['post_idx'].cuda()

    def run():
        net.zero_grad()
        embed = net(c['graph'], c['node_id'].cuda(), c['etype'].cuda(), c['enorm'].cuda())
        lg = net.get_loss(c['graph'], embed, c['samples'].cuda(), c['labels'].cuda())
        lg[0].backward()
        return embed, lg

    with ops.gemm_precision('bf16'):
        embed, lg = run()
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    close(embed, enc_b['z'], rtol=5e-3, atol_scale=5e-3, msg='z (bf16 operands)')
    for a, b, nm in zip(lg, lo_b, ('loss', 'pred', 'kl', 'mmd')):
        close(a, b, rtol=5e-3, atol_scale=5e-3, msg=nm + ' (bf16 operands)')
    for name, g in grads.items():
        if st_b[name].grad is not None:
            close(g, st_b[name].grad, rtol=2e-2, atol_scale=2e-2, msg='grad ' + name + ' (bf16 operands)')
    # the mode is really on (differs from fp32) and stays near the fp32 result
    zf = enc_f['z'].detach()
    dz = float((embed.detach().cpu() - zf).abs().max()) / float(zf.abs().max())
    assert 1e-5 < dz < 5e-2, dz
    l_b, l_f = float(lg[0].detach()), float(lo_f[0].detach())
    assert abs(l_b - l_f) < 5e-2 * max(1.0, abs(l_f))
    # and the same modules in fp32 meet the 1e-4 bar on this shape (10x10 / 10x20 blocks, R = 22)
    embed32, lg32 = run()
    close(embed32, enc_f['z'], msg='z fp32')
    for a, b, nm in zip(lg32, lo_f, ('loss', 'pred', 'kl', 'mmd')):
        close(a, b, msg=nm + ' fp32')
    for name, p in net.named_parameters():
        if st_f[name].grad is not None:
            close(p.grad, st_f[name].grad, rtol=5e-4, atol_scale=5e-5, msg='grad ' + name + ' fp32')


@pytest.mark.parametrize('device_sampler', [False, True])
def test_train_driver_end_to_end_with_checkpoint_round_trip(tmp_path, device_sampler, capsys):
    """gcn_vae_amd.train.main with the reference's flags (kgvae/link_predict.py:272-322): a few mini-batch steps, the
    periodic validation (fused raw-MRR ranker) with its checkpoint, then --test-mode from that checkpoint
    ({'state_dict', 'epoch'}, kgvae/link_predict.py:245-259) -- SURVEY 8(f-4)."""
    from gcn_vae_amd import train
    ckpt = str(tmp_path / 'model_state.pth')
    argv = ['-d', 'synthetic:400:9:3000:150:150:1', '--gpu', '0', '--n-hidden', '16', '--n-bases', '4', '--n-epochs', '6',
            '--evaluate-every', '3', '--graph-batch-size', '600', '--eval-batch-size', '50', '--mmd-param', '1.0',
            '--n-flows', '2', '--mog-k', '4', '--model-state-file', ckpt]
    if device_sampler:
        argv.append('--device-sampler')
    np.random.seed(0)
    random.seed(0)
    torch.manual_seed(0)
    best = train.main(train.build_parser().parse_args(argv))
    out = capsys.readouterr().out
    assert out.count('Epoch 00') == 6 and out.count('start eval') == 2 and 'training done' in out
    assert 0.0 < best <= 1.0
    saved = torch.load(ckpt, map_location='cpu')
    assert set(saved) == {'state_dict', 'epoch'} and saved['epoch'] in (3, 6)
    keys = set(saved['state_dict'])
    assert {'w_relation', 'encoder.z_pre', 'encoder.pi', 'encoder.input_layer.embedding.weight',
            'encoder.rconv_layer_1.weight', 'encoder.rconv_layer_2.loop_weight', 'encoder.nf.0.net.0.mask',
            'encoder.nf.2.net.0.weight'} <= keys
    args = train.build_parser().parse_args(argv)
    args.test_mode = True
    mrr = train.main(args)
    assert 0.0 < mrr <= 1.0
    assert 'Using best epoch' in capsys.readouterr().out


def test_identity_embedding_gradient_accumulates_and_output_does_not_alias_for_other_callers():
    """Full-graph training looks the embedding table up with ids == arange: the encoder's layer 1 then stores dL/dx rows
    straight into the table's gradient -- only while that buffer is known to be zero.  Two backward() calls before one
    step() must ADD (gradient accumulation), a gradient already written by another lookup must survive, and a lookup by
    anyone but the fused layer returns a copy, never a view of the parameter."""
    from gcn_vae_amd import ops, sampling
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.optim import FlatAdam
    from gcn_vae_amd.train import LinkPredict
    n, n_rel, h = 400, 8, 16
    data = synthetic_kg(n, n_rel, 3000, seed=0)
    g, rel, node_norm = sampling.build_test_graph(n, n_rel, data.train)
    _, dst = g.edges()
    torch.manual_seed(0)
    net = LinkPredict(KGVAE, n, h, n_rel, num_bases=4, num_hidden_layers=2, dropout=0.0, use_cuda=True, reg_param=0.01,
                      kl_param=1e-3, mmd_param=0.0, k=4, n_flows=0).cuda().train()
    opt = FlatAdam([p for p in net.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0)
    node_id = torch.arange(n, device='cuda').view(-1, 1)
    et = torch.from_numpy(rel).cuda()
    enorm = torch.from_numpy(node_norm).cuda()[dst.cuda()].view(-1, 1).contiguous()
    np.random.seed(0)
    samples, labels = sampling.negative_sampling(data.train[:500], n, 3)
    trip, lab = torch.from_numpy(samples).cuda(), torch.from_numpy(labels).cuda()
    net.encoder.eps_override = torch.randn(n, h, generator=torch.Generator().manual_seed(1)).cuda()
    table = net.encoder.input_layer.embedding.weight

    def backward_once():
        embed = net(g, node_id, et, enorm)
        net.get_loss(g, embed, trip, lab)[0].backward()

    opt.zero_grad()
    backward_once()
    one = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.requires_grad}
    assert float(one['encoder.input_layer.embedding.weight'].abs().max()) > 0
    backward_once()                                   # no zero_grad in between: every gradient doubles, the table's too
    for k, p in net.named_parameters():
        if p.requires_grad:
            torch.testing.assert_close(p.grad, 2 * one[k], rtol=1e-5, atol=1e-7 * float(one[k].abs().max() + 1e-30), msg=k)
    # a gradient that another consumer left in the table's buffer first is added to, not erased
    opt.zero_grad()
    table.grad.fill_(0.25)
    ops.GRAD_FRESH.discard(table.grad.data_ptr())
    backward_once()
    torch.testing.assert_close(table.grad, one['encoder.input_layer.embedding.weight'] + 0.25, rtol=1e-5, atol=1e-6)
    # lookups outside the fused layer get a copy
    out = ops.embedding(table, node_id.view(-1))
    assert out.data_ptr() != table.data_ptr()
    before = table.detach().clone()
    out.detach().mul_(0.0)
    assert torch.equal(table.detach(), before)


def _minibatch_net(data, h=32, n_flows=0, seed=0):
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.train import LinkPredict
    import itertools
    from gcn_vae_amd import ops
    ops._rngs.clear()                            # a fresh device generator (tick 0) and the same stream ids for every net built
    ops._rng_streams = itertools.count(1)        # here: two nets of one test then draw the same dropout masks and noise
    torch.manual_seed(seed)
    return LinkPredict(KGVAE, data.num_nodes, h, data.num_rels, num_bases=8, num_hidden_layers=2, dropout=0.2, use_cuda=True,
                       reg_param=0.01, kl_param=1e-3, mmd_param=1.0, k=10, n_flows=n_flows).cuda().train()


@pytest.mark.parametrize('sampler', ['uniform', 'neighbor'])
def test_static_shape_sampler_equals_the_synchronising_one(sampler):
    """DeviceSampler.sample_static (no host sync, node arrays padded to cap, batch number read from device memory) builds the
    batch that DeviceSampler.sample builds for the same seed and batch number: same triplets, same graph, same negatives;
    the node count is on the device, a padding row's node id is its own position."""
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.device_sampling import DeviceSampler
    data = synthetic_kg(3000, 11, 20000, seed=2)
    a = DeviceSampler(data.train, data.num_nodes, data.num_rels, 'cuda', seed=5, sampler=sampler)
    b = DeviceSampler(data.train, data.num_nodes, data.num_rels, 'cuda', seed=5, sampler=sampler)
    pick = torch.zeros(200, dtype=torch.int64, device='cuda')
    for _ in range(3):
        dyn, sta = a.sample(900, 0.5, 4), b.sample_static(900, 0.5, 4, mmd_pick=pick)
        n = int(sta.rows_dev.item())
        cap = min(1800, data.num_nodes)
        assert n == dyn.node_id.shape[0] and sta.node_id.shape == (cap, 1) and sta.g.number_of_nodes() == cap
        assert torch.equal(sta.node_id[:n], dyn.node_id)
        assert torch.equal(sta.node_id[n:].view(-1), torch.arange(n, cap, device='cuda'))        # padding: distinct valid ids
        assert torch.equal(sta.samples, dyn.samples) and torch.equal(sta.labels, dyn.labels)
        assert torch.equal(sta.edge_type, dyn.edge_type) and torch.equal(sta.edge_norm, dyn.edge_norm)
        for x, y in zip(sta.g.edges(), dyn.g.edges()):
            assert torch.equal(x, y)
        p = pick.cpu().numpy()
        assert len(np.unique(p)) == 200 and p.min() >= 0 and p.max() < n           # distinct existing rows


def test_loss_head_with_padding_rows_equals_the_loss_on_the_existing_rows():
    """include/gcnvae.h rows_dev: z padded with garbage rows + the device row count gives the loss and the gradients of the
    unpadded call (padding rows: zero gradient)."""
    from gcn_vae_amd import ops
    gen = torch.Generator().manual_seed(0)
    n, cap, h, k, T, R = 700, 1000, 64, 10, 5000, 9
    z = torch.randn(cap, h, generator=gen)
    m, v = torch.randn(cap, h, generator=gen), torch.rand(cap, h, generator=gen) + 0.1
    w_rel, z_pre = torch.randn(R, h, generator=gen) * 0.3, torch.randn(2 * k, h, generator=gen) * 0.5
    z_pri = torch.randn(200, h, generator=gen)
    pick = torch.randperm(n, generator=gen)[:200]
    trip = torch.stack([torch.randint(0, n, (T,), generator=gen), torch.randint(0, R, (T,), generator=gen),
                        torch.randint(0, n, (T,), generator=gen)], 1)
    labels = (torch.rand(T, generator=gen) > 0.5).float()

    def run(rows, rows_dev):
        ins = [t[:rows].clone().cuda().requires_grad_(True) for t in (z, m, v)] + \
              [t.clone().cuda().requires_grad_(True) for t in (w_rel, z_pre, z_pri)]
        tidx = ops.TripletIndex(trip.cuda(), rows, R, sync_free=True)
        out = ops.loss_head(ins[0], ins[1], ins[2], ins[3], ins[4], None, ins[5], pick.cuda(), labels.cuda(), tidx, 0.01, 1e-2, 1.0,
                            False, rows_dev=rows_dev)
        out[0].backward()
        return [o.detach().cpu() for o in out], [t.grad.cpu() for t in ins]
    (la, pa, ka, ma), ga = run(n, None)
    (lb, pb, kb, mb), gb = run(cap, torch.tensor([n], dtype=torch.int32, device='cuda'))
    for x, y, nm in ((la, lb, 'loss'), (pa, pb, 'pred'), (ka, kb, 'kl'), (ma, mb, 'mmd')):
        torch.testing.assert_close(y.reshape(-1), x.reshape(-1), rtol=1e-5, atol=1e-6, msg=lambda s, nm=nm: f'{nm}: {s}')
    for i, nm in enumerate(('z', 'z_mean', 'z_sigma')):
        torch.testing.assert_close(gb[i][:n], ga[i], rtol=1e-4, atol=1e-7, msg=lambda s, nm=nm: f'grad {nm}: {s}')
        assert float(gb[i][n:].abs().max()) == 0.0, nm
    for i, nm in ((3, 'w_rel'), (4, 'z_pre'), (5, 'z_pri')):
        torch.testing.assert_close(gb[i], ga[i], rtol=1e-4, atol=1e-7, msg=lambda s, nm=nm: f'grad {nm}: {s}')


@pytest.mark.parametrize('n_flows', [0, 2])
def test_graphed_minibatch_step_equals_eager_steps(n_flows):
    """kgvae/link_predict.py:200-236 as one hipGraph (gcn_vae_amd.graph_step): six training steps -- three eager warm-up steps of
    the static-shape body, then three replays -- follow the same losses as six eager steps with the synchronising sampler
    (dynamic shapes, host-side MMD row pick replaced by the same device draw), and end at the same parameters."""
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.device_sampling import STREAM_PICK, DeviceSampler
    from gcn_vae_amd.graph_step import GraphedMiniBatchStep
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    from gcn_vae_amd.optim import FlatAdam
    data = synthetic_kg(3000, 11, 20000, seed=2)
    k, split, neg = 1500, 0.5, 5
    # --- eager, dynamic shapes
    net_e = _minibatch_net(data, n_flows=n_flows)
    opt_e = FlatAdam(net_e.parameters(), lr=1e-2, max_grad_norm=1.0)
    sm_e = DeviceSampler(data.train, data.num_nodes, data.num_rels, 'cuda', seed=9)
    pick_e = torch.zeros(200, dtype=torch.int64, device='cuda')
    net_e.encoder.mmd_index_override = pick_e
    losses_e = []
    for _ in range(6):
        b = sm_e.sample(k, split, neg)
        n = b.node_id.shape[0]
        p32 = torch.empty(200, dtype=torch.int32, device='cuda')
        lib.call('gv_perm_sample', n, 200, sm_e.seed, sm_e.tick, None, None, STREAM_PICK, ptr(p32), lib.stream())
        pick_e.copy_(p32)
        opt_e.zero_grad()
        embed = net_e(b.g, b.node_id, b.edge_type, b.edge_norm)
        out = net_e.get_loss(b.g, embed, b.samples, b.labels)
        out[0].backward()
        opt_e.step()
        losses_e.append([float(t.detach()) for t in out])
    # --- captured
    net_g = _minibatch_net(data, n_flows=n_flows)
    opt_g = FlatAdam(net_g.parameters(), lr=1e-2, max_grad_norm=1.0)
    sm_g = DeviceSampler(data.train, data.num_nodes, data.num_rels, 'cuda', seed=9)
    step = GraphedMiniBatchStep(net_g, opt_g, sm_g, k, split, neg)
    step.capture(warmup=3)               # three eager steps of the static-shape body, then the recording (nothing runs in it)
    losses_g = [[float(t.detach()) for t in step()] for _ in range(3)]
    # steps 4, 5, 6 of both runs
    for e, g in zip(losses_e[3:], losses_g):
        np.testing.assert_allclose(g, e, rtol=2e-3, atol=1e-5)
    for (kk, pe), (_, pg) in zip(net_e.named_parameters(), net_g.named_parameters()):
        torch.testing.assert_close(pg, pe, rtol=2e-2, atol=2e-3, msg=lambda s, kk=kk: f'{kk}: {s}')
    assert losses_g[0] != losses_g[1] != losses_g[2]                       # every replay draws a fresh batch


def test_fused_reparam_kl_equals_the_separate_nodes(monkeypatch):
    """ops.reparam(h2, eps, z_pre) + loss_head (K3 fused into K6: gv_reparam_kl_fwd / gv_reparam_kl_bwd) against the separate
    reparameterisation and KL passes on the same inputs: loss terms and every gradient, incl. dL/dh2 and the mixture's."""
    from gcn_vae_amd import ops
    gen = torch.Generator().manual_seed(3)
    n, h, k, T, R = 900, 72, 10, 4000, 7
    h2 = torch.randn(n, 2 * h, generator=gen)
    h2[0, h:h + 4] = torch.tensor([25.0, -30.0, 19.99, 20.01])           # softplus threshold on both sides
    eps = torch.randn(n, h, generator=gen)
    w_rel, z_pre = torch.randn(R, h, generator=gen) * 0.3, torch.randn(2 * k, h, generator=gen) * 0.5
    z_pri = torch.randn(200, h, generator=gen)
    pick = torch.randperm(n, generator=gen)[:200].cuda()
    trip = torch.stack([torch.randint(0, n, (T,), generator=gen), torch.randint(0, R, (T,), generator=gen),
                        torch.randint(0, n, (T,), generator=gen)], 1).cuda()
    labels = (torch.rand(T, generator=gen) > 0.5).float().cuda()

    def run(fused):
        monkeypatch.setattr(ops, 'FUSE_REPARAM_KL', fused)
        ins = [t.clone().cuda().requires_grad_(True) for t in (h2, w_rel, z_pre, z_pri)]
        z, m, v = ops.reparam(ins[0], eps.cuda(), ins[2])
        assert (getattr(z, '_gv_kl_link', None) is not None) == fused
        tidx = ops.TripletIndex(trip, n, R, sync_free=True)
        out = ops.loss_head(z, m, v, ins[1], ins[2], None, ins[3], pick, labels, tidx, 0.01, 1e-2, 1.0, False)
        out[0].backward()
        return [o.detach().cpu() for o in out], [t.grad.cpu() for t in ins], (z.detach().cpu(), m.detach().cpu(), v.detach().cpu())
    oa, ga, za = run(False)
    ob, gb, zb = run(True)
    for x, y in zip(za, zb):
        assert torch.equal(x, y)                                       # same arithmetic, same bits
    for x, y, nm in zip(oa, ob, ('loss', 'pred', 'kl', 'mmd')):
        torch.testing.assert_close(y.reshape(-1), x.reshape(-1), rtol=1e-6, atol=1e-7, msg=lambda s, nm=nm: f'{nm}: {s}')
    for x, y, nm in zip(ga, gb, ('h2', 'w_rel', 'z_pre', 'z_pri')):
        torch.testing.assert_close(y, x, rtol=1e-5, atol=1e-8, msg=lambda s, nm=nm: f'grad {nm}: {s}')


def test_direct_gradient_registry_change_between_forward_and_backward_is_refused():
    """INTEGRATION.md, the optimiser contract: a backward whose forward resolved a gradient-arena target refuses to run after
    the registry changed (a FlatAdam built or dropped in between) instead of adding into an arena nobody reads."""
    from gcn_vae_amd import ops
    from gcn_vae_amd.optim import FlatAdam
    w = torch.nn.Parameter(torch.randn(12, 8, device='cuda'))
    b = torch.nn.Parameter(torch.randn(12, device='cuda'))
    table = torch.nn.Parameter(torch.randn(30, 8, device='cuda'))
    opt = FlatAdam([w, b, table], lr=1e-3)
    ids = torch.tensor([3, 5, 5, 7], device='cuda').view(-1, 1)
    out = ops.embedding(table, ids)
    y = ops.linear(out, w, b, 0).sum()
    opt2 = FlatAdam([torch.nn.Parameter(torch.randn(4, device='cuda'))], lr=1e-3)       # registry changes here
    with pytest.raises(RuntimeError, match='gradient arena'):
        y.backward()
    del opt2
    opt.zero_grad()
    y2 = ops.linear(ops.embedding(table, ids), w, b, 0).sum()
    y2.backward()                                                                       # an undisturbed pair works
    assert float(table.grad.abs().sum()) > 0 and table.grad.data_ptr() == opt.flat_g[opt.offsets[table]:].data_ptr()


@pytest.mark.parametrize('bf16', [False, True])
def test_train_driver_with_graph_step(tmp_path, capsys, bf16):
    """gcn_vae_amd.train.main --device-sampler --graph-step: every step is one hipGraph replay; evaluation and the checkpoint
    round trip work as in the eager loop.  With --bf16 the flows run on the one-launch-per-pass MADE kernels inside the graph."""
    from gcn_vae_amd import ops, train
    ckpt = str(tmp_path / 'model_state.pth')
    argv = ['-d', 'synthetic:400:9:3000:150:150:1', '--gpu', '0', '--n-hidden', '16', '--n-bases', '4', '--n-epochs', '8',
            '--evaluate-every', '4', '--graph-batch-size', '600', '--eval-batch-size', '50', '--mmd-param', '1.0', '--kl-param', '1e-3',
            '--n-flows', '2', '--mog-k', '4', '--model-state-file', ckpt, '--device-sampler', '--graph-step'] + (['--bf16'] if bf16 else [])
    torch.manual_seed(0)
    try:
        best = train.main(train.build_parser().parse_args(argv))
    finally:
        ops.set_gemm_precision('f32')
    out = capsys.readouterr().out
    assert out.count('Epoch 00') == 8 and 'training done' in out and 0.0 < best <= 1.0       # 3 eager steps (ordinary epochs) + 5 replays
    assert 'Mean step time' in out and out.count('start eval') == 2
    losses = [float(line.split('Loss ')[1].split(' |')[0]) for line in out.splitlines() if line.startswith('Epoch 00')]
    assert all(np.isfinite(losses)) and len(set(losses)) == len(losses)


@pytest.mark.gpu
@pytest.mark.parametrize('blocks', [2, 3])
def test_made_passes_over_row_blocks_on_their_own_streams_give_the_same_bits(monkeypatch, blocks):
    """made._by_row_blocks: the bf16 MADE node runs its passes over independent row blocks on side streams (every launch of a pass
    is row-local) -- outputs and every gradient equal the one-block run bit for bit, also when the block boundary is not the end
    of the rows' last 64-row tile and with the weight gradients going through the optimiser arena (the side-stream products)."""
    from gcn_vae_amd import made, ops
    from gcn_vae_amd.flows import MADE
    from gcn_vae_amd.optim import FlatAdam
    n, d = 1000, 40                       # 16 row tiles, the last one partial
    z = torch.randn(n, d, generator=torch.Generator().manual_seed(5)).cuda()
    res, seen = [], []
    inner = made._by_row_blocks
    monkeypatch.setattr(made, '_by_row_blocks', lambda run, rows, want, *a, **k: (seen.append((rows, want is None)), inner(run, rows, want, *a, **k))[1])
    for k in (1, blocks):
        monkeypatch.setattr(made, 'MADE_ROW_BLOCKS', k)
        monkeypatch.setattr(made, 'MADE_ROW_BLOCKS_MIN_TILES', 1)
        torch.manual_seed(3)
        m = MADE(d, 56, 2).cuda()
        opt = FlatAdam(list(m.parameters()), lr=1e-3, max_grad_norm=1.0)
        opt.zero_grad()
        with ops.gemm_precision('bf16'):
            zz = z.clone().requires_grad_(True)
            x, ld = m(zz)
            (x.sin().sum() + (ld * ld).sum()).backward()
        torch.cuda.synchronize()
        res.append((x.detach().clone(), ld.detach().clone(), zz.grad.clone(), [p.grad.detach().clone() for p in m.parameters()]))
        opt.close()
    assert len(made._made_row_blocks(n)) == blocks and made._made_row_blocks(n)[1][0] % 64 == 0
    assert seen and all(t for _, t in seen)            # the node took the path with the tiled copies (the one that is split)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert float(res[0][2].abs().max()) > 0
    for a, b in zip(res[0][3], res[1][3]):
        assert torch.equal(a, b) and float(a.abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('n,d,hidden,n_hidden', [(1000, 40, 56, 2), (64, 200, 200, 3), (4300, 200, 200, 3), (130, 8, 16, 1)])
def test_iaf_update_backward_as_the_first_stage_of_the_backward_chain_gives_the_same_bits(monkeypatch, n, d, hidden, n_hidden):
    """gv_made_chain_iafb: the bf16 MADE node's backward with the IAF update's backward made in the prologue of the backward chain
    ([g_mu | g_alpha] straight into layer 0's LDS tile, the transposed copy through a 64-column LDS block, g_z added in place)
    against the two-launch form (gv_iaf_update_bwd_bf16_ex, then gv_made_chain): x, log-det, dL/dz and every parameter gradient
    bit for bit -- with and without a log-det gradient, row counts that end inside a 64-row tile, one tile, d = 8 (one column
    block, partly filled) and d = 200 (four blocks, the last one 8 columns).  The one-launch form with ONE launch per pass, with
    groups of two passes (n_passes = 2: the last group may hold one) and with all passes of the node in one launch (n_passes up to
    6: a pass reads the fp32 output its workgroup wrote in the pass before)."""
    from gcn_vae_amd import made, ops
    from gcn_vae_amd.flows import MADE
    z = torch.randn(n, d, generator=torch.Generator().manual_seed(n + d)).cuda()
    calls = []
    inner = made.made_chain
    monkeypatch.setattr(made, 'made_chain', lambda x, *a, **k: (calls.append(k.get('stage') is not None), inner(x, *a, **k))[1])
    for with_ld in (True, False):
        res = []
        for on, per_launch in ((False, 1), (True, 1), (True, 2), (True, 6)):
            calls.clear()
            monkeypatch.setattr(made, 'MADE_CHAIN_IAFB', on)
            monkeypatch.setattr(made, 'MADE_CHAIN_PASSES', per_launch)
            torch.manual_seed(3)
            m = MADE(d, hidden, n_hidden).cuda()
            with ops.gemm_precision('bf16'):
                zz = z.clone().requires_grad_(True)
                x, ld = m(zz)
                (x.sin().sum() + ((ld * ld).sum() if with_ld else 0.0)).backward()
            torch.cuda.synchronize()
            assert any(calls) == on, 'the node did not take the expected backward form'
            if on:      # P = n_hidden + 3 index sets: P - 1 passes in groups of per_launch, per row block
                groups = -(-(n_hidden + 2) // per_launch)
                assert sum(calls) == groups * len(made._made_row_blocks(n)), (sum(calls), groups)
            res.append((x.detach().clone(), ld.detach().clone(), zz.grad.clone(), [p.grad.detach().clone() for p in m.parameters()]))
        assert float(res[0][2].abs().max()) > 0
        for other in res[1:]:
            assert torch.equal(res[0][0], other[0]) and torch.equal(res[0][1], other[1]) and torch.equal(res[0][2], other[2])
            for a, b in zip(res[0][3], other[3]):
                assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('n,d,hidden,n_hidden', [(1000, 40, 56, 2), (64, 200, 200, 3), (4300, 200, 200, 3), (130, 8, 16, 1), (700, 64, 96, 4)])
def test_all_forward_passes_of_a_made_in_one_launch_give_the_same_bits(monkeypatch, n, d, hidden, n_hidden):
    """gv_made_chain_fwd: the bf16 MADE node's forward passes 1 .. P-1 in ONE launch (a workgroup keeps its 64 rows, x_new stays
    in LDS as the next pass's input; the buffer roles swap from pass to pass when the layer count is odd: n_hidden + 2 layers)
    against one gv_made_chain launch per pass: x, log-det, dL/dz and every parameter gradient -- i.e. everything the forward
    stores for the backward: exp(alpha + mu), the sign words and the tiled transposed copies of every pass -- bit for bit; in
    groups of two passes and with all passes in one launch; row counts that end inside a 64-row tile (the last workgroup's rows
    past m leave as zeros in the tiled copies and store nothing else)."""
    from gcn_vae_amd import made, ops
    from gcn_vae_amd.flows import MADE
    z = torch.randn(n, d, generator=torch.Generator().manual_seed(n + d)).cuda()
    calls = []
    inner = made.made_chain_fwd
    monkeypatch.setattr(made, 'made_chain_fwd', lambda x, m, layers, passes, **k: (calls.append(len(passes)), inner(x, m, layers, passes, **k))[1])
    res = []
    for per_launch in (1, 2, 6, 0):         # (0: chosen by size -- all passes where the rows run as one block, three otherwise)
        calls.clear()
        monkeypatch.setattr(made, 'MADE_FWD_PASSES', per_launch)
        per_launch = per_launch or (6 if len(made._made_row_blocks(n)) == 1 else 3)
        torch.manual_seed(3)
        m = MADE(d, hidden, n_hidden).cuda()
        with ops.gemm_precision('bf16'):
            zz = z.clone().requires_grad_(True)
            x, ld = m(zz)
            (x.sin().sum() + (ld * ld).sum()).backward()
        torch.cuda.synchronize()
        if per_launch == 1:
            assert not calls
        else:       # P - 1 = n_hidden + 2 passes in groups of per_launch, per row block
            assert sum(calls) == (n_hidden + 2) * len(made._made_row_blocks(n)) and max(calls) == min(per_launch, n_hidden + 2), calls
        res.append((x.detach().clone(), ld.detach().clone(), zz.grad.clone(), [p.grad.detach().clone() for p in m.parameters()]))
    assert float(res[0][2].abs().max()) > 0 and bool(torch.isfinite(res[0][0]).all())
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0]) and torch.equal(res[0][1], other[1]) and torch.equal(res[0][2], other[2])
        for a, b in zip(res[0][3], other[3]):
            assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('n,d,hidden,n_hidden', [(1000, 40, 56, 2), (4300, 200, 200, 3), (130, 8, 16, 1)])
def test_permute_layer_folded_into_the_bf16_made_node_gives_the_same_bits(monkeypatch, n, d, hidden, n_hidden):
    """MADE.forward(reverse_out=True) on the bf16 node -- the PermuteLayer behind an IAF block (kgvae/model.py:60-66): the last
    forward pass stores x with its columns reversed (gv_chain_fwd_pass.flags), the first backward launch reads dL/dx and the
    handed-through gradient that way (gv_chain_iafb.flags bit 1) -- against the node followed by ops.reverse_cols: x, log-det,
    dL/dz, every parameter gradient bit for bit; also where the passes go one launch each (the reversal then stays a launch)."""
    from gcn_vae_amd import made, ops
    from gcn_vae_amd.flows import MADE
    z = torch.randn(n, d, generator=torch.Generator().manual_seed(n + d)).cuda()
    wgt = torch.randn(n, d, generator=torch.Generator().manual_seed(1)).cuda()       # (a loss that tells the columns apart)
    rev_calls = []
    inner = made.lib.call
    monkeypatch.setattr(made.lib, 'call', lambda name, *a, **k: (rev_calls.append(name) if name == 'gv_reverse_cols' else None, inner(name, *a, **k))[1])
    res = []
    for fold, per_launch in ((False, 0), (True, 0), (True, 1)):
        rev_calls.clear()
        monkeypatch.setattr(made, 'MADE_FWD_PASSES', per_launch)
        torch.manual_seed(3)
        m = MADE(d, hidden, n_hidden).cuda()
        with ops.gemm_precision('bf16'):
            zz = z.clone().requires_grad_(True)
            if fold:
                x, ld = m(zz, reverse_out=True)
            else:
                x, ld = m(zz)
                x = ops.reverse_cols(x)
            ((x * wgt).sin().sum() + (ld * ld).sum()).backward()
        torch.cuda.synchronize()
        if fold and per_launch == 0:
            assert not rev_calls, 'the reversal was expected to ride along'
        res.append((x.detach().clone(), ld.detach().clone(), zz.grad.clone(), [p.grad.detach().clone() for p in m.parameters()]))
    assert float(res[0][2].abs().max()) > 0
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0]) and torch.equal(res[0][1], other[1]) and torch.equal(res[0][2], other[2])
        for a, b in zip(res[0][3], other[3]):
            assert torch.equal(a, b)


@pytest.mark.gpu
def test_forward_passes_launch_refuses_what_it_cannot_run():
    """gv_made_chain_fwd's argument checks (include/gcnvae.h): more than six passes, a hidden layer without ReLU / without its tiled
    copy, a pass without its sign words or IAF operands -- a status and a message, no launch; the well-formed call runs."""
    from gcn_vae_amd import made
    m, d, h = 100, 16, 32
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(h, d, generator=g).cuda() * 0.1, torch.randn(2 * d, h, generator=g).cuda() * 0.1]
    (pf0, _), (pf1, _) = made.made_pack_weights(ws, iaf_last=True)
    x = torch.randn(m, d, generator=g).cuda().to(torch.bfloat16)
    z, xo = torch.randn(m, d, generator=g).cuda(), torch.randn(m, d, generator=g).cuda()
    cc = torch.ones(d, dtype=torch.int32, device=dev)
    T, tt = (m + 63) // 64, 64 * max(d, h)
    f32, bf = dict(dtype=torch.float32, device=dev), dict(dtype=torch.bfloat16, device=dev)

    def pass_():
        return dict(x_old=xo, colcount=cc, ex=torch.empty(m, d, **f32), x_new=torch.empty(m, d, **f32),
                    out_bf16_t=torch.zeros(T * tt, **bf), act_t=[torch.zeros(T * tt, **bf)],
                    act_bits=[torch.zeros(m, 1, dtype=torch.int32, device=dev)])

    def layers(relu=True, tiled=True):
        first = dict(w_packed=pf0, n=h, k=d, relu=relu, out_bits=torch.zeros(m, 1, dtype=torch.int32, device=dev))
        if tiled:
            first.update(out_bf16_t=torch.zeros(T * tt, **bf), t_tile=tt)
        return [first, dict(w_packed=pf1, n=2 * d, k=h, iaf=dict(z=z, x_old=xo, colcount=cc, ex=torch.empty(m, d, **f32)),
                            out_bf16=torch.empty(m, d, **bf), out_bf16_t=torch.zeros(T * tt, **bf), t_tile=tt)]

    made.made_chain_fwd(x, m, layers(), [pass_(), pass_()])
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='n_passes'):
        made.made_chain_fwd(x, m, layers(), [pass_() for _ in range(7)])
    with pytest.raises(RuntimeError, match='hidden layer 0'):
        made.made_chain_fwd(x, m, layers(relu=False), [pass_()])
    with pytest.raises(RuntimeError, match='hidden layer 0'):
        made.made_chain_fwd(x, m, layers(tiled=False), [pass_()])
    bad = pass_()
    bad['act_bits'] = [None]
    with pytest.raises(RuntimeError, match='sign words'):
        made.made_chain_fwd(x, m, layers(), [bad])
    bad = pass_()
    bad['ex'] = None
    with pytest.raises((RuntimeError, KeyError), match='pass 0|ex'):
        made.made_chain_fwd(x, m, layers(), [bad])
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize('precision,n', [('bf16', 9000), ('f32', 5000)])
def test_multi_stream_flow_stack_equals_the_plain_one_eagerly_and_in_a_captured_step(precision, n):
    """tools/probes/made_stress.py: two MADE blocks in a row with FlatAdam -- two row blocks on their own streams, weight gradients
    on the side stream (the first block's products beside the second block's backward), prepared parameters -- against the same
    stack with all of that off: every output and gradient of several eager steps and of several replays of ONE captured step,
    bit for bit."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('made_stress', os.path.join(root, 'tools', 'probes', 'made_stress.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.check(n=n, steps=3, precision=precision, d=64)


@pytest.mark.gpu
def test_flow_parameter_work_prepared_beside_the_encoder_gives_the_same_bits(monkeypatch):
    """ops.made_prepare (KGVAE.forward announces its MADE calls: mask folds, packed weights and pass 0's row run on a side stream
    beside the R-GCN layers and are picked up behind an event) against the node doing that work itself: same embedding, same loss,
    same gradients, bit for bit; nothing prepared is left over."""
    from gcn_vae_amd import made, ops, sampling
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.train import LinkPredict
    n, n_rel, h = 600, 8, 16
    data = synthetic_kg(n, n_rel, 4000, seed=1)
    g, rel, node_norm = sampling.build_test_graph(n, n_rel, data.train)
    _, dst = g.edges()
    node_id = torch.arange(n, device='cuda').view(-1, 1)
    et = torch.from_numpy(rel).cuda()
    enorm = torch.from_numpy(node_norm).cuda()[dst.cuda()].view(-1, 1).contiguous()
    np.random.seed(0)
    samples, labels = sampling.negative_sampling(data.train[:500], n, 3)
    trip, lab = torch.from_numpy(samples).cuda(), torch.from_numpy(labels).cuda()
    eps = torch.randn(n, h, generator=torch.Generator().manual_seed(1)).cuda()
    res, used = [], []
    inner = made._made_params_work
    monkeypatch.setattr(made, '_made_params_work', lambda *a: (used.append(torch.cuda.current_stream().cuda_stream), inner(*a))[1])
    for on in (False, True):
        monkeypatch.setattr(made, 'MADE_PREPARE', on)
        torch.manual_seed(0)
        net = LinkPredict(KGVAE, n, h, n_rel, num_bases=4, num_hidden_layers=2, dropout=0.0, use_cuda=True, reg_param=0.01,
                          kl_param=1e-3, mmd_param=0.0, k=4, n_flows=2).cuda().train()
        net.encoder.eps_override = eps
        main = torch.cuda.current_stream().cuda_stream
        del used[:]
        with ops.gemm_precision('bf16'):
            embed = net(g, node_id, et, enorm)
            loss = net.get_loss(g, embed, trip, lab)[0]
            loss.backward()
        torch.cuda.synchronize()
        assert len(used) == 2 and all((st != main) == on for st in used), (on, used, main)       # one call per flow, on the side stream when prepared
        assert not made._made_prep
        res.append((embed.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[0][2].keys() == res[1][2].keys() and any('nf' in k for k in res[0][2])
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k


@pytest.mark.gpu
def test_row_blocks_kept_forked_over_the_flow_stack_give_the_same_bits(monkeypatch):
    """made.keep_row_blocks_forked (KGVAE._apply_flows): the bf16 MADE nodes of a three-block IAF stack run their row blocks on side
    streams that stay forked from the first node to the last -- one fork, one join, the log-det row sums behind it, every node's
    temporaries held until then -- against a fork and a join per node: embedding, loss and every gradient of two
    consecutive steps bit for bit (the captured step: bench.py --config c3, whose parity leg runs it against the oracle)."""
    from gcn_vae_amd import made, ops, sampling
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.train import LinkPredict
    n, n_rel, h = 1000, 8, 16
    data = synthetic_kg(n, n_rel, 6000, seed=1)
    g, rel, node_norm = sampling.build_test_graph(n, n_rel, data.train)
    _, dst = g.edges()
    node_id = torch.arange(n, device='cuda').view(-1, 1)
    et = torch.from_numpy(rel).cuda()
    enorm = torch.from_numpy(node_norm).cuda()[dst.cuda()].view(-1, 1).contiguous()
    np.random.seed(0)
    samples, labels = sampling.negative_sampling(data.train[:500], n, 3)
    trip, lab = torch.from_numpy(samples).cuda(), torch.from_numpy(labels).cuda()
    eps = torch.randn(n, h, generator=torch.Generator().manual_seed(1)).cuda()
    monkeypatch.setattr(made, 'MADE_ROW_BLOCKS', 2)
    monkeypatch.setattr(made, 'MADE_ROW_BLOCKS_MIN_TILES', 1)
    joins = []
    inner = made._fork_join
    monkeypatch.setattr(made, '_fork_join', lambda st: (joins.append(st['open']), inner(st))[1])
    res = []
    for on in (False, True):
        monkeypatch.setattr(made, 'MADE_KEEP_FORKED', on)
        torch.manual_seed(0)
        net = LinkPredict(KGVAE, n, h, n_rel, num_bases=4, num_hidden_layers=2, dropout=0.0, use_cuda=True, reg_param=0.01,
                          kl_param=1e-3, mmd_param=0.0, k=4, n_flows=3).cuda().train()
        net.encoder.eps_override = eps
        del joins[:]
        outs = []
        with ops.gemm_precision('bf16'):
            for _ in range(2):
                net.zero_grad(set_to_none=True)
                embed = net(g, node_id, et, enorm)
                loss = net.get_loss(g, embed, trip, lab)[0]
                loss.backward()
                torch.cuda.synchronize()
                outs.append((embed.detach().clone(), loss.detach().clone(),
                             {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
        assert (joins == [True, True]) if on else not joins, joins          # one join of the kept fork per forward pass
        assert made._FORK is None and not made._made_prep
        res.append(outs)
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2].keys() == b[2].keys() and any('nf' in k for k in a[2])
        for k in a[2]:
            assert torch.equal(a[2][k], b[2][k]), k


@pytest.mark.gpu
def test_rgcn_weight_gradients_on_the_backward_side_stream_give_the_same_bits(monkeypatch):
    """ops.rgcn_bwd_side: with FlatAdam registered the R-GCN layers' block-weight and self-loop-weight gradients are stored into the
    arena on the backward side stream, beside the dL/dx path (joined when the backward pass has run): every gradient and the
    weights after a clipped Adam step bit for bit as with everything on one stream; 'auto' takes the side stream for a large layer
    alone and not behind IAF blocks (their weight-gradient products run there)."""
    from gcn_vae_amd import ops, sampling
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.optim import FlatAdam
    from gcn_vae_amd.train import LinkPredict
    n, n_rel, h = 500, 8, 16
    data = synthetic_kg(n, n_rel, 4000, seed=0)
    g, rel, node_norm = sampling.build_test_graph(n, n_rel, data.train)
    _, dst = g.edges()
    node_id = torch.arange(n, device='cuda').view(-1, 1)
    et = torch.from_numpy(rel).cuda()
    enorm = torch.from_numpy(node_norm).cuda()[dst.cuda()].view(-1, 1).contiguous()
    np.random.seed(0)
    samples, labels = sampling.negative_sampling(data.train[:500], n, 3)
    trip, lab = torch.from_numpy(samples).cuda(), torch.from_numpy(labels).cuda()
    eps = torch.randn(n, h, generator=torch.Generator().manual_seed(1)).cuda()
    taken = []
    inner = ops.backward_side

    def spy(enabled, *held, **kw):
        if kw.get('rgcn'):
            taken.append(True)
        return inner(enabled, *held, **kw)
    monkeypatch.setattr(ops, 'backward_side', spy)
    res = []
    for mode, flows, work, expect in (('0', 0, 1, 0), ('1', 0, 1, 2), ('auto', 0, 1, 2), ('auto', 0, 10 ** 12, 0), ('auto', 2, 1, 0), ('0', 2, 1, 0)):
        monkeypatch.setattr(ops, 'RGCN_BWD_SIDE', mode)
        monkeypatch.setattr(ops, 'RGCN_BWD_SIDE_MIN_WORK', work)
        del taken[:]
        torch.manual_seed(0)
        net = LinkPredict(KGVAE, n, h, n_rel, num_bases=4, num_hidden_layers=2, dropout=0.0, use_cuda=True, reg_param=0.01,
                          kl_param=1e-3, mmd_param=0.0, k=4, n_flows=flows).cuda().train()
        net.encoder.eps_override = eps
        opt = FlatAdam([p for p in net.parameters() if p.requires_grad], lr=1e-2, max_grad_norm=1.0)
        opt.zero_grad()
        embed = net(g, node_id, et, enorm)
        net.get_loss(g, embed, trip, lab)[0].backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        opt.step()
        torch.cuda.synchronize()
        assert len(taken) == expect, (mode, flows, work, taken)        # both R-GCN layers, or neither
        res.append((flows, grads, {k: p.detach().clone() for k, p in net.named_parameters()}))
        opt.close()
    for flows, grads, weights in res[1:]:
        ref = res[0] if flows == 0 else res[-1]
        assert grads.keys() == ref[1].keys() and any('loop_weight' in k for k in grads)
        for k in grads:
            assert torch.equal(grads[k], ref[1][k]), k
        for k in weights:
            assert torch.equal(weights[k], ref[2][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['bf16', 'f32'])
def test_made_gradients_written_straight_into_the_optimiser_arena_equal_autograd(precision):
    """With FlatAdam registered, the MADE nodes' (bf16 and fp32) masked weight gradients and bias gradients are stored straight into
    the (all-zero) arena slices -- on the side stream that is joined when the backward pass has run -- instead of going through
    AccumulateGrad: same numbers as plain autograd, and a second backward() without zero_grad() in between doubles them (the
    slices are then no longer 'fresh': accumulate)."""
    from gcn_vae_amd import ops, sampling
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.optim import FlatAdam
    from gcn_vae_amd.train import LinkPredict
    n, n_rel, h = 400, 8, 16
    data = synthetic_kg(n, n_rel, 3000, seed=0)
    g, rel, node_norm = sampling.build_test_graph(n, n_rel, data.train)
    _, dst = g.edges()
    node_id = torch.arange(n, device='cuda').view(-1, 1)
    et = torch.from_numpy(rel).cuda()
    enorm = torch.from_numpy(node_norm).cuda()[dst.cuda()].view(-1, 1).contiguous()
    np.random.seed(0)
    samples, labels = sampling.negative_sampling(data.train[:500], n, 3)
    trip, lab = torch.from_numpy(samples).cuda(), torch.from_numpy(labels).cuda()
    eps = torch.randn(n, h, generator=torch.Generator().manual_seed(1)).cuda()

    def build():
        torch.manual_seed(0)
        net = LinkPredict(KGVAE, n, h, n_rel, num_bases=4, num_hidden_layers=2, dropout=0.0, use_cuda=True, reg_param=0.01,
                          kl_param=1e-3, mmd_param=0.0, k=4, n_flows=2).cuda().train()
        net.encoder.eps_override = eps
        return net

    def backward_once(net):
        embed = net(g, node_id, et, enorm)
        net.get_loss(g, embed, trip, lab)[0].backward()

    with ops.gemm_precision(precision):
        ref = build()
        backward_once(ref)
        want = {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None}
        net = build()
        opt = FlatAdam([p for p in net.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0)
        opt.zero_grad()
        backward_once(net)
        flow_keys = [k for k in want if 'flow' in k.lower() or 'made' in k.lower() or '.net.' in k]
        assert flow_keys, sorted(want)
        for k, p in net.named_parameters():
            if k in want:
                torch.testing.assert_close(p.grad, want[k], rtol=1e-5, atol=1e-7 * float(want[k].abs().max() + 1e-30), msg=k)
        backward_once(net)
        for k, p in net.named_parameters():
            if k in want:
                torch.testing.assert_close(p.grad, 2 * want[k], rtol=1e-5, atol=1e-7 * float(want[k].abs().max() + 1e-30), msg=k)
        del opt


@pytest.mark.parametrize('precision,n', [('bf16', 3000), ('f32', 3000)])
def test_two_made_nodes_on_shared_weights_order_their_arena_writes(monkeypatch, precision, n):
    """ONE MADE's parameters behind TWO autograd nodes of the same backward pass (KGVAE's posterior pass and a
    separate MMD prior pass).  The node that runs first stores its weight gradients into the fresh arena slices on the 'bwd' side
    stream; the second one sees the slices no longer fresh and accumulates on the main stream -- ops.backward_side makes that
    stream wait for the side stream first.  Held to the same pass with the side stream off, bit for bit (both orders of writes are
    then the same sequence of stores and adds), also when backward() is called under another stream than the forward ran on."""
    from gcn_vae_amd import ops
    from gcn_vae_amd.flows import MADE
    from gcn_vae_amd.optim import FlatAdam
    d = 64
    gen = torch.Generator().manual_seed(11)
    z1, z2 = torch.randn(n, d, generator=gen).cuda(), torch.randn(200, d, generator=gen).cuda()
    res = []
    other = torch.cuda.Stream()
    for side_on, other_stream in ((False, False), (True, False), (True, True)):
        monkeypatch.setattr(ops, 'BWD_SIDE', side_on)
        torch.manual_seed(3)
        m = MADE(d, d, 2).cuda()
        opt = FlatAdam(list(m.parameters()), lr=1e-3, max_grad_norm=1.0)
        opt.zero_grad()
        with ops.gemm_precision(precision):
            x1, ld1 = m(z1)
            x2, ld2 = m(z2)
            loss = x1.sin().sum() + (ld1 * ld1).sum() + x2.cos().sum() + ld2.sum()
            if other_stream:
                other.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(other):
                    loss.backward()
                    opt.step()          # reads the arena on the caller's stream: the join has to cover it
                torch.cuda.current_stream().wait_stream(other)
            else:
                loss.backward()
                opt.step()
        torch.cuda.synchronize()
        assert not ops._bwd_side_held
        res.append([p.detach().clone() for p in m.parameters()])
        opt.close()
    for a, b, c in zip(*res):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b) and torch.equal(a, c)


def test_sharded_flat_adam_on_one_rank_equals_flat_adam():
    """distributed.ShardedFlatAdam without a process group (one piece = the whole arena) against optim.FlatAdam on the same
    gradients: identical bits while the clip is inactive (the update is element-wise), 1e-6 when it is active (the two sum the
    squares in different orders); the gradient arena is consumed; snapshot / restore go through the piece's moments."""
    from gcn_vae_amd import distributed as gdist
    from gcn_vae_amd.optim import FlatAdam
    gen = torch.Generator().manual_seed(3)
    shapes = [(37, 16), (16,), (200, 8), (5,)]
    res = []
    for make in (lambda ps: FlatAdam(ps, lr=1e-2, max_grad_norm=1.0), lambda ps: gdist.ShardedFlatAdam(ps, lr=1e-2, max_grad_norm=1.0)):
        g2 = torch.Generator().manual_seed(4)
        params = [torch.nn.Parameter(torch.randn(*sh, generator=g2).cuda()) for sh in shapes]
        opt = make(params)
        snap, outs = None, []
        for step, scale in enumerate((1e-3, 1e-3, 5.0)):          # two steps under the clip threshold, one far above it
            gs = torch.Generator().manual_seed(10 + step)
            opt.zero_grad()
            for p_ in params:
                p_.grad.copy_(torch.randn(*p_.shape, generator=gs).cuda() * scale)
            opt.step()
            torch.cuda.synchronize()
            assert float(opt.flat_g.abs().max()) == 0.0
            outs.append([p_.detach().clone() for p_ in params])
            if step == 0:
                snap = opt.snapshot()
        opt.restore(snap)
        outs.append([p_.detach().clone() for p_ in params])
        res.append(outs)
        opt.close()
    for step in (0, 1):
        for a, b in zip(res[0][step], res[1][step]):
            assert torch.equal(a, b)
    for a, b in zip(res[0][2], res[1][2]):
        torch.testing.assert_close(b, a, rtol=1e-6, atol=1e-7)
    for a, b, c in zip(res[1][3], res[1][0], res[0][3]):
        assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize('n_flows', [0, 2])
def test_reference_validation_block_runs_unchanged_through_compat(tmp_path, n_flows):
    """kgvae/link_predict.py:239-261 replayed line for line on the aliases compat.install() registers: the reference moves its
    model to the CPU for validation (``model.cpu()``), forwards HOST tensors, ranks with ``utils.calc_mrr`` and moves back.
    The scorer class below is the reference's own Python in shape (an nn.Module owning ``w_relation`` and the encoder), so
    ``.cpu()`` really moves ITS parameter; the HIP modules keep theirs on the device, copy the host inputs over and compute
    there -- same embedding and same MRR as with everything on the GPU.  No ``--evaluate-every`` workaround needed."""
    import sys
    import torch.nn as nn
    from gcn_vae_amd import compat
    saved = {k: sys.modules.get(k) for k in ('dgl', 'dgl.nn', 'dgl.nn.pytorch', 'dgl.contrib', 'dgl.contrib.data', 'model',
                                             'flow_network', 'utils')}
    try:
        compat.install()
        import utils
        from model import KGVAE

        class LinkPredict(nn.Module):          # kgvae/link_predict.py:30-63, constructor and forward
            def __init__(self, in_dim, h_dim, num_rels):
                super().__init__()
                self.encoder = KGVAE(in_dim, h_dim, h_dim, num_rels * 2, 4, 2, 0.2, True, True, k=3, n_flows=n_flows)
                self.w_relation = nn.Parameter(torch.Tensor(num_rels, h_dim))
                nn.init.xavier_uniform_(self.w_relation, gain=nn.init.calculate_gain('relu'))

            def forward(self, g, h, r, norm):
                return self.encoder.forward(g, h, r, norm)

        from gcn_vae_amd.data import synthetic_kg
        data = synthetic_kg(400, 7, 3000, seed=2)
        num_nodes, num_rels = data.num_nodes, data.num_rels
        torch.manual_seed(0)
        model = LinkPredict(num_nodes, 16, num_rels)
        valid_data = torch.LongTensor(data.train[:300])        # (the synthetic set has no validation split)
        val_graph, val_rel, val_norm = utils.build_test_graph(num_nodes, num_rels, data.train)
        val_deg = val_graph.in_degrees(range(val_graph.number_of_nodes())).float().view(-1, 1)
        val_node_id = torch.arange(0, num_nodes, dtype=torch.long).view(-1, 1)
        val_rel = torch.from_numpy(val_rel)
        val_norm = utils.node_norm_to_edge_norm(val_graph, torch.from_numpy(val_norm).view(-1, 1))
        use_cuda = True
        model.cuda()
        model.encoder.eps_override = torch.randn(num_nodes, 16, device='cuda')      # the reparameterisation draws noise in eval mode too: pin it
        # everything on the GPU: the result the block below has to reproduce
        model.eval()
        with torch.no_grad():
            embed_gpu = model(val_graph, val_node_id.cuda(), val_rel.cuda(), val_norm.cuda())
            mrr_gpu = utils.calc_mrr(embed_gpu, model.w_relation, valid_data, hits=[1, 3, 10], eval_bz=100, all_batches=False,
                                     flow_log_prob=model.encoder.get_flow_log_prob() if n_flows else None, verbose=False)
        # ---- kgvae/link_predict.py:239-261 ----
        if use_cuda:
            model.cpu()
        model.eval()
        state_file = str(tmp_path / 'model_state.pth')
        torch.save({'state_dict': model.state_dict(), 'epoch': 1}, state_file)
        embed = model(val_graph, val_node_id, val_rel, val_norm)
        mrr = utils.calc_mrr(embed, model.w_relation, valid_data, hits=[1, 3, 10], eval_bz=100, all_batches=False,
                             flow_log_prob=model.encoder.get_flow_log_prob() if n_flows else None, verbose=False)
        if use_cuda:
            model.cuda()
        # ----
        assert embed.is_cuda and all(p.is_cuda for p in model.parameters())
        assert torch.equal(embed.detach(), embed_gpu) and mrr == mrr_gpu and 0.0 < mrr <= 1.0
        ck = torch.load(state_file)
        assert set(ck['state_dict']) == set(model.state_dict()) and ck['epoch'] == 1
        # the reference's own parameter did move to the host in between (the HIP modules' did not)
        model.cpu()
        assert not model.w_relation.is_cuda and all(p.is_cuda for p in model.encoder.parameters())
        model.cuda()
        # a training step still works after the round trip
        model.train()
        z = model(val_graph, val_node_id.cuda(), val_rel.cuda(), val_norm.cuda())
        (z.sum() + model.encoder.get_kl(z)).backward()
        assert model.encoder.rconv_layer_1.weight.grad is not None
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
