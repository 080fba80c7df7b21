"""The oracle against the golden vectors captured from the reference (tests/golden/make_golden.py).

CPU only.  Tolerances: the oracle restates the same torch ops in the same order, so outputs are
compared at 1e-6 relative (bit-exact in the generating container); integer outputs exactly."""
import random

import numpy as np
import torch

import seeded
from conftest import load_golden
from oracle import flows, graphs, kgvae, prob, ranking

TOL = dict(rtol=1e-6, atol=1e-6)


def test_made_masks_degrees_and_outputs():
    g = load_golden('made.npz')
    for tag, (d, h, nh, seed) in {'d16': (16, 16, 3, 100), 'd200': (200, 200, 3, 200), 'd8h12': (8, 12, 2, 300)}.items():
        layers = seeded.made_layers(seed, d, h, nh)
        z = g[f'{tag}_z']
        x, ld = flows.made_forward(z, layers, d, h, nh)
        torch.testing.assert_close(x, g[f'{tag}_x'], **TOL)
        torch.testing.assert_close(ld, g[f'{tag}_logdet'], **TOL)
        zi, ldi = flows.made_inverse(z, layers, d, h, nh)
        torch.testing.assert_close(zi, g[f'{tag}_inv'], **TOL)
        torch.testing.assert_close(ldi, g[f'{tag}_inv_logdet'], **TOL)
        if d <= 16:
            for i, m in enumerate(flows.made_masks(d, h, nh)):
                assert torch.equal(m, g[f'{tag}_mask{i}'])
            for i, idx in enumerate(flows.made_degrees(d, h, nh)):
                assert torch.equal(idx, g[f'{tag}_m{i}'])
        # backward wiring
        lay = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in layers]
        zg = z.clone().requires_grad_(True)
        xg, ldg = flows.made_forward(zg, lay, d, h, nh)
        (xg.pow(2).sum() + ldg.sum()).backward()
        torch.testing.assert_close(zg.grad, g[f'{tag}_grad_z'], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(lay[0][0].grad, g[f'{tag}_grad_w0'], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(lay[-1][0].grad, g[f'{tag}_grad_wlast'], rtol=1e-5, atol=1e-5)
    px, pld = flows.permute(g['perm_in'])
    assert torch.equal(px, g['perm_out']) and torch.equal(pld, g['perm_logdet'])


def test_made_forward_is_not_inverted_by_inverse():
    # documents the reference's behaviour, guards against "fixing" it
    layers = seeded.made_layers(100, 16, 16, 3)
    z = seeded.randn(150, 8, 16, scale=0.5)
    x, _ = flows.made_forward(z, layers, 16, 16, 3)
    back, _ = flows.made_inverse(x, layers, 16, 16, 3)
    assert (back - z).abs().max() > 1e-2


def test_probability_helpers():
    g = load_golden('prob.npz')
    m, v = prob.gaussian_parameters(g['gp_in'])
    torch.testing.assert_close(m, g['gp_m'], **TOL)
    torch.testing.assert_close(v, g['gp_v'], **TOL)
    m1, v1 = prob.gaussian_parameters(g['gp1_in'], dim=1)
    m0, v0 = prob.gaussian_parameters(g['gp1_in'].squeeze(0), dim=0)
    for a, b in ((m1, 'gp1_m'), (v1, 'gp1_v'), (m0, 'gp0_m'), (v0, 'gp0_v')):
        torch.testing.assert_close(a, g[b], **TOL)
    edge = torch.cat([g['gp_edge_in'], g['gp_edge_in']], 1)
    torch.testing.assert_close(prob.gaussian_parameters(edge)[1], g['gp_edge_v'], **TOL)
    torch.testing.assert_close(prob.sample_gaussian(m, v, g['sg_eps']), g['sg_out'], **TOL)
    torch.testing.assert_close(prob.sample_gaussian(m1, v1, g['sg_rep_eps'], repeat=3), g['sg_rep_out'], **TOL)
    torch.testing.assert_close(prob.log_normal(g['ln_x'], g['ln_m'], g['ln_v']), g['ln_out'], **TOL)
    torch.testing.assert_close(prob.log_normal_mixture(g['ln_x'], m1, v1), g['lnm_out'], **TOL)
    torch.testing.assert_close(prob.log_sum_exp(g['lse_in'], 0), g['lse_d0'], **TOL)
    torch.testing.assert_close(prob.log_sum_exp(g['lse_in'], 1), g['lse_d1'], **TOL)
    torch.testing.assert_close(prob.log_mean_exp(g['lse_in'], 1), g['lme_d1'], **TOL)


def test_graph_pipeline_matches_reference_rng_stream():
    g = load_golden('pipeline.npz')
    train = g['train'].numpy()
    adj, deg = graphs.get_adj_and_degrees(300, train)
    assert np.array_equal(deg, g['degrees'].numpy())
    assert np.array_equal(np.concatenate([a.reshape(-1, 2) for a in adj if a.size]), g['adj_flat'].numpy())
    np.random.seed(0)
    ns, nl = graphs.negative_sampling(g['neg_pos'].numpy().copy(), 300, 3)
    assert np.array_equal(ns, g['neg_samples'].numpy()) and np.array_equal(nl, g['neg_labels'].numpy())
    np.random.seed(1)
    assert np.array_equal(graphs.sample_edge_uniform(adj, deg, len(train), 100), g['uniform_edges'].numpy())
    np.random.seed(2)
    assert np.array_equal(graphs.sample_edge_neighborhood(adj, deg, len(train), 60), g['neighbor_edges'].numpy())
    for tag, sampler, seed in (('u', 'uniform', 3), ('n', 'neighbor', 4)):
        np.random.seed(seed)
        gr, uniq_v, rel, norm, samples, labels = graphs.generate_sampled_graph_and_labels(
            train, 200, 0.5, 12, adj, deg, 4, sampler)
        src, dst = gr.edges()
        assert torch.equal(src, g[f'{tag}_src']) and torch.equal(dst, g[f'{tag}_dst'])
        assert np.array_equal(uniq_v, g[f'{tag}_uniq_v'].numpy())
        assert np.array_equal(rel, g[f'{tag}_rel'].numpy())
        assert np.array_equal(norm, g[f'{tag}_norm'].numpy())
        assert np.array_equal(samples, g[f'{tag}_samples'].numpy())
        assert np.array_equal(labels, g[f'{tag}_labels'].numpy())
        en = graphs.node_norm_to_edge_norm(gr, torch.from_numpy(norm).view(-1, 1))
        assert torch.equal(en, g[f'{tag}_edge_norm'])
        # structural properties of the reference's edge order
        key = dst * (1 << 40) + src * (1 << 20) + torch.from_numpy(rel)
        assert bool((key[1:] >= key[:-1]).all())
    tg, trel, tnorm = graphs.build_test_graph(300, 12, g['valid'].numpy())
    ts, td = tg.edges()
    assert torch.equal(ts, g['test_src']) and torch.equal(td, g['test_dst'])
    assert np.array_equal(trel, g['test_rel'].numpy()) and np.array_equal(tnorm, g['test_norm'].numpy())


def test_ranking():
    g = load_golden('ranking.npz')
    trip = g['trip']
    rs = ranking.perturb_and_get_rank(g['emb'], g['w'], trip[:, 2], trip[:, 1], trip[:, 0], 37, 10, True, g['flp'])
    assert torch.equal(rs, g['ranks_s'])
    mrr, hits, _ = ranking.calc_mrr(g['emb'], g['w'], trip, hits=[1, 3, 10], eval_bz=10, all_batches=True,
                                    flow_log_prob=g['flp'])
    assert abs(mrr - float(g['mrr'])) < 1e-7
    mrr1, _, _ = ranking.calc_mrr(g['emb'], g['w'], trip, hits=[1], eval_bz=10, all_batches=False,
                                  flow_log_prob=g['flp'])
    assert abs(mrr1 - float(g['mrr_first_batch'])) < 1e-7


def _run_model(tag, n_flows, kl, mmd):
    g = load_golden(f'model_c1_{tag}.npz')
    state = {k[6:]: v.clone().requires_grad_(v.is_floating_point() and not k.endswith('mask') and not k.endswith('.pi'))
             for k, v in g.items() if k.startswith('state.')}
    enc = kgvae.kgvae_encode(state, g['src'], g['dst'], g['node_id'].view(-1, 1), g['etype'], g['edge_norm'],
                             g['eps'], 4, n_flows)
    loss, pred, klv, mmdv = kgvae.link_predict_loss(state, enc, g['samples'], g['labels'], 0.01, kl, mmd, 10,
                                                    n_flows, g['eps_prior'], g['post_idx'])
    return g, state, enc, (loss, pred, klv, mmdv)


def test_whole_model_c1_with_flows():
    g, state, enc, (loss, pred, kl, mmd) = _run_model('flows3', 3, 1e-5, 1.0)
    torch.testing.assert_close(enc['z'], g['z'], **TOL)
    torch.testing.assert_close(enc['z_mean'], g['z_mean'], **TOL)
    torch.testing.assert_close(enc['z_sigma'], g['z_sigma'], **TOL)
    torch.testing.assert_close(enc['flow_log_prob'], g['flow_log_prob'], **TOL)
    for a, b in ((loss, 'loss'), (pred, 'pred'), (kl, 'kl'), (mmd, 'mmd')):
        torch.testing.assert_close(a.reshape(()), g[b].reshape(()), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(kgvae.distmult_score(enc['z'], state['w_relation'], g['samples']), g['score'], **TOL)
    loss.backward()
    n_checked = 0
    for k, v in g.items():
        if k.startswith('grad.'):
            torch.testing.assert_close(state[k[5:]].grad, v, rtol=1e-5, atol=1e-6, msg=k)
            n_checked += 1
    assert n_checked >= 6 + 30   # w_relation, z_pre, embedding, 2x3 layer params, 3x5x2 flow params


def test_whole_model_c1_without_flows():
    g, state, enc, (loss, pred, kl, mmd) = _run_model('flows0', 0, 0.0, 1.0)
    assert enc['flow_log_prob'] is None
    torch.testing.assert_close(enc['z'], g['z'], **TOL)
    for a, b in ((loss, 'loss'), (pred, 'pred'), (mmd, 'mmd')):
        torch.testing.assert_close(a.reshape(()), g[b].reshape(()), rtol=1e-6, atol=1e-6)
    loss.backward()
    for k, v in g.items():
        if k.startswith('grad.'):
            torch.testing.assert_close(state[k[5:]].grad, v, rtol=1e-5, atol=1e-6, msg=k)


def test_philox_known_answers():
    """Random123's published known-answer vectors for philox4x32_10 (kat_vectors: zeros, ones, digits of pi)."""
    import numpy as np
    from oracle import philox

    def run(ctr, key):
        return [int(v) for v in philox.philox4x32_10(key, np.array([ctr], dtype=np.uint32))[0]]

    assert run([0, 0, 0, 0], (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run([0xffffffff] * 4, (0xffffffff, 0xffffffff)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_sampler_draw_restatement_known_answers():
    """oracle/philox.py's restatement of the batch sampler's draws (gv_perm_sample, gv_negative_sampling): structural
    properties and pinned values (the GPU tests compare the kernels against these functions bit for bit)."""
    from oracle import philox
    x = philox.perm_sample(272115, 20000, 123, 5, 0x5A01)
    assert x[:8].tolist() == [97718, 268098, 203867, 234506, 222920, 146317, 253150, 53913]
    assert len(set(x.tolist())) == 20000 and x.min() >= 0 and x.max() < 272115
    for n in (1, 2, 3, 64, 65, 1000):
        full = philox.perm_sample(n, n, 9, 1, 7)
        assert sorted(full.tolist()) == list(range(n))
    assert not np.array_equal(philox.perm_sample(1000, 1000, 9, 1, 7), philox.perm_sample(1000, 1000, 9, 2, 7))
    v, h = philox.negative_draws(100000, 10212, 1, 2, 3)
    assert v.min() >= 0 and v.max() < 10212 and 0.49 < h.mean() < 0.51
    assert np.bincount(v * 10 // 10212, minlength=10).min() > 9500          # uniform over the entities
