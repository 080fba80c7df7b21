#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference; the GPU box never runs this).
The reference's Python is imported from where it lies, under the ``dgl`` stand-in of
``oracle/dgl_shim.py``; only inputs and outputs are written --
no reference source text.  Everything except the RelGraphConv layer is the reference's
own code; the layer inside the whole-model fixture is ``oracle.rgcn`` (parity unpinned).

    python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch

from oracle import dgl_shim, graphs, kgvae as okg  # noqa: E402
import seeded  # noqa: E402

dgl_shim.install()
sys.path.insert(0, '/root/reference/kgvae')
import flow_network as ref_flow  # noqa: E402
import utils as ref_utils  # noqa: E402
import model as ref_model  # noqa: E402
import link_predict as ref_lp  # noqa: E402

torch.autograd.set_detect_anomaly(False)   # the reference switches it on at import (model.py:10)


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print(f'wrote {name}: {os.path.getsize(path) / 1024:.1f} kB')


# ---------------------------------------------------------------- (i) MADE
def gen_made():
    out = {}
    for tag, (d, h, nh, n, seed) in {'d16': (16, 16, 3, 8, 100), 'd200': (200, 200, 3, 8, 200),
                                     'd8h12': (8, 12, 2, 5, 300)}.items():
        m = ref_flow.MADE(d, h, nh)
        layers = seeded.made_layers(seed, d, h, nh)
        with torch.no_grad():
            for i, (w, b) in enumerate(layers):
                m.net[2 * i].weight.copy_(w)
                m.net[2 * i].bias.copy_(b)
        z = seeded.randn(seed + 50, n, d, scale=0.5)
        x, ld = m.forward(z)
        zi, ldi = m.inverse(z)
        out.update({f'{tag}_z': z, f'{tag}_x': x, f'{tag}_logdet': ld, f'{tag}_inv': zi, f'{tag}_inv_logdet': ldi})
        if d <= 16:
            for i in range(nh + 2):
                out[f'{tag}_mask{i}'] = m.net[2 * i].mask
            for i, idx in enumerate(m.m):
                out[f'{tag}_m{i}'] = idx
        # gradient of a scalar functional through forward (pins the backward wiring)
        zg = z.clone().requires_grad_(True)
        xg, ldg = m.forward(zg)
        (xg.pow(2).sum() + ldg.sum()).backward()
        out[f'{tag}_grad_z'] = zg.grad
        out[f'{tag}_grad_w0'] = m.net[0].weight.grad
        out[f'{tag}_grad_wlast'] = m.net[2 * (nh + 1)].weight.grad
    p = ref_flow.PermuteLayer(7)
    t = seeded.randn(9, 3, 7)
    px, pld = p.forward(t)
    out.update(perm_in=t, perm_out=px, perm_logdet=pld)
    save('made.npz', **out)


# ---------------------------------------------------------------- (ii)+(iii) probability helpers
def gen_prob():
    out = {}
    h = seeded.randn(1, 6, 10)
    m, v = ref_utils.gaussian_parameters(h)
    out.update(gp_in=h, gp_m=m, gp_v=v)
    zp = seeded.randn(2, 1, 8, 5)                       # like z_pre (1, 2k, h)
    m1, v1 = ref_utils.gaussian_parameters(zp, dim=1)
    m0, v0 = ref_utils.gaussian_parameters(zp.squeeze(0), dim=0)
    out.update(gp1_in=zp, gp1_m=m1, gp1_v=v1, gp0_m=m0, gp0_v=v0)
    big = torch.tensor([[-30.0, -5.0, 0.0, 5.0, 19.9, 20.1, 50.0, 1e-3]])
    out.update(gp_edge_in=big, gp_edge_v=ref_utils.gaussian_parameters(torch.cat([big, big], 1))[1])
    torch.manual_seed(11)
    s = ref_utils.sample_gaussian(m, v)
    torch.manual_seed(11)
    eps = torch.randn(*m.shape)
    out.update(sg_eps=eps, sg_out=s)
    torch.manual_seed(12)
    s_rep = ref_utils.sample_gaussian(m1, v1, repeat=3)
    torch.manual_seed(12)
    eps_rep = torch.randn(12, 5)
    out.update(sg_rep_eps=eps_rep, sg_rep_out=s_rep)
    x = seeded.randn(3, 7, 5)
    out.update(ln_x=x, ln_m=m1[0, :1].expand(7, 5), ln_v=v1[0, :1].expand(7, 5))
    out['ln_out'] = ref_utils.log_normal(x, out['ln_m'], out['ln_v'])
    out['lnm_out'] = ref_utils.log_normal_mixture(x, m1, v1)
    y = seeded.randn(4, 6, 9, scale=3.0)
    out.update(lse_in=y, lse_d0=ref_utils.log_sum_exp(y, 0), lse_d1=ref_utils.log_sum_exp(y, 1),
               lme_d1=ref_utils.log_mean_exp(y, 1))
    save('prob.npz', **out)


# ---------------------------------------------------------------- (v) numpy graph pipeline
def gen_pipeline():
    data = dgl_shim.SyntheticKG(300, 12, 2500, 200, 200, seed=5)
    out = dict(train=data.train)
    adj, deg = ref_utils.get_adj_and_degrees(data.num_nodes, data.train)
    out['degrees'] = deg
    out['adj_flat'] = np.concatenate([a.reshape(-1, 2) for a in adj if a.size])
    np.random.seed(0)
    pos = data.train[:40].copy()
    ns, nl = ref_utils.negative_sampling(pos, 300, 3)
    out.update(neg_pos=pos, neg_samples=ns, neg_labels=nl)
    np.random.seed(1)
    out['uniform_edges'] = ref_utils.sample_edge_uniform(adj, deg, len(data.train), 100)
    np.random.seed(2)
    out['neighbor_edges'] = ref_utils.sample_edge_neighborhood(adj, deg, len(data.train), 60)
    for tag, sampler, seed in (('u', 'uniform', 3), ('n', 'neighbor', 4)):
        np.random.seed(seed)
        g, uniq_v, rel, norm, samples, labels = ref_utils.generate_sampled_graph_and_labels(
            data.train, 200, 0.5, data.num_rels, adj, deg, 4, sampler)
        src, dst = g.edges()
        enorm = ref_lp.node_norm_to_edge_norm(g, torch.from_numpy(norm).view(-1, 1))
        out.update({f'{tag}_src': src, f'{tag}_dst': dst, f'{tag}_uniq_v': uniq_v, f'{tag}_rel': rel,
                    f'{tag}_norm': norm, f'{tag}_samples': samples, f'{tag}_labels': labels,
                    f'{tag}_edge_norm': enorm})
    tg, trel, tnorm = ref_utils.build_test_graph(data.num_nodes, data.num_rels, torch.LongTensor(data.valid))
    ts, td = tg.edges()
    out.update(valid=data.valid, test_src=ts, test_dst=td, test_rel=trel, test_norm=tnorm)
    save('pipeline.npz', **out)


# ---------------------------------------------------------------- (vi) ranking
def gen_ranking():
    emb = seeded.randn(21, 60, 8, scale=0.7)
    w = seeded.randn(22, 5, 8, scale=0.7)
    rs = np.random.RandomState(23)
    trip = torch.from_numpy(np.stack([rs.randint(0, 60, 37), rs.randint(0, 5, 37), rs.randint(0, 60, 37)], 1))
    flp = torch.tensor(-0.3)
    ranks_s = ref_utils.perturb_and_get_rank(emb, w, trip[:, 2], trip[:, 1], trip[:, 0], 37, 10, True, flp)
    mrr = ref_utils.calc_mrr(emb, w, trip, hits=[1, 3, 10], eval_bz=10, all_batches=True, flow_log_prob=flp)
    mrr1 = ref_utils.calc_mrr(emb, w, trip, hits=[1, 3, 10], eval_bz=10, all_batches=False, flow_log_prob=flp)
    save('ranking.npz', emb=emb, w=w, trip=trip, flp=flp, ranks_s=ranks_s, mrr=np.float64(mrr),
         mrr_first_batch=np.float64(mrr1))


# ---------------------------------------------------------------- (iv)+(vii)+(viii) whole model, C1 size
def gen_model():
    data = dgl_shim.SyntheticKG(1000, 20, 6000, 300, 300, seed=0)
    adj, deg = ref_utils.get_adj_and_degrees(data.num_nodes, data.train)
    np.random.seed(0)
    g, node_id, etype, node_norm, samples, labels = ref_utils.generate_sampled_graph_and_labels(
        data.train, 2000, 0.5, data.num_rels, adj, deg, 10, 'uniform')
    node_id_t = torch.from_numpy(node_id).view(-1, 1).long()
    etype_t = torch.from_numpy(etype)
    enorm = ref_lp.node_norm_to_edge_norm(g, torch.from_numpy(node_norm).view(-1, 1))
    samples_t, labels_t = torch.from_numpy(samples), torch.from_numpy(labels)
    src, dst = g.edges()
    n = len(node_id)
    base = dict(src=src, dst=dst, node_id=node_id, etype=etype, edge_norm=enorm, samples=samples,
                labels=labels)
    manifest = {}
    for tag, cfg in {'flows3': dict(n_flows=3, kl=1e-5, mmd=1.0, k=10),
                     'flows0': dict(n_flows=0, kl=0.0, mmd=1.0, k=10)}.items():
        torch.manual_seed(0)
        net = ref_lp.LinkPredict(ref_model.KGVAE, data.num_nodes, 16, data.num_rels, num_bases=4,
                                 num_hidden_layers=2, dropout=0.0, use_cuda=False, reg_param=0.01,
                                 kl_param=cfg['kl'], mmd_param=cfg['mmd'], k=cfg['k'], n_flows=cfg['n_flows'])
        with torch.no_grad():                      # non-trivial biases so the bias path is exercised
            net.encoder.rconv_layer_1.h_bias.copy_(seeded.randn(61, 16, scale=0.1))
            net.encoder.rconv_layer_2.h_bias.copy_(seeded.randn(62, 32, scale=0.1))
        net.train()
        torch.manual_seed(123)
        random.seed(7)
        embed = net(g, node_id_t, etype_t, enorm)
        loss, pred, kl, mmd = net.get_loss(g, embed, samples_t, labels_t)
        loss.backward()
        torch.manual_seed(123)
        eps = torch.randn(n, 16)
        eps_prior = torch.randn(200, 16)
        random.seed(7)
        post_idx = np.array(random.sample(range(n), 200))
        state = {k: v.detach().clone() for k, v in net.state_dict().items()}
        # cross-check: the oracle reproduces the reference here (RelGraphConv is the oracle's on both sides)
        enc = okg.kgvae_encode(state, src, dst, node_id_t, etype_t, enorm, eps, 4, cfg['n_flows'])
        ol = okg.link_predict_loss(state, enc, samples_t, labels_t, 0.01, cfg['kl'], cfg['mmd'], cfg['k'],
                                   cfg['n_flows'], eps_prior, torch.from_numpy(post_idx))
        print(tag, 'oracle-vs-reference |dz|', (enc['z'] - embed).abs().max().item(),
              '|dloss|', abs(ol[0].item() - loss.item()), 'loss', loss.item(), pred.item(), kl.item(), mmd.item())
        out = dict(base)
        out.update({'state.' + k: v for k, v in state.items()})
        out.update({'grad.' + k: p.grad for k, p in net.named_parameters() if p.grad is not None})
        flp = net.encoder.get_flow_log_prob()
        out.update(eps=eps, eps_prior=eps_prior, post_idx=post_idx, z=embed, z_mean=net.encoder.z_mean,
                   z_sigma=net.encoder.z_sigma, loss=loss, pred=pred, kl=kl, mmd=mmd,
                   score=net.calc_score(embed, samples_t), reg=net.regularization_loss(embed))
        if flp is not None:
            out['flow_log_prob'] = flp
        save(f'model_c1_{tag}.npz', **out)
        manifest['LinkPredict(KGVAE,n_flows=%d)' % cfg['n_flows']] = {k: list(v.shape) for k, v in state.items()}
        manifest['grads_missing_' + tag] = [k for k, p in net.named_parameters() if p.grad is None]
    torch.manual_seed(0)
    rg = ref_model.RGCN(50, 8, 8, 6, 2, num_hidden_layers=2, dropout=0.0, use_self_loop=True, use_cuda=False)
    manifest['RGCN(num_hidden_layers=2)'] = {k: list(v.shape) for k, v in rg.state_dict().items()}
    with open(os.path.join(HERE, 'state_dict_manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print('wrote state_dict_manifest.json')


if __name__ == '__main__':
    gen_made()
    gen_prob()
    gen_pipeline()
    gen_ranking()
    gen_model()
