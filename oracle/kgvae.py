"""Oracle restatement of KGVAE / RGCN / LinkPredict  --  TEST INFRASTRUCTURE.

Wiring, KL, MMD, scorer and loss are pinned by golden vectors generated from the
reference (``tests/golden/make_golden.py``); the R-GCN layer inside is ``oracle.rgcn``
(parity unpinned, see its header).

Follows
  /root/reference/kgvae/model.py:107-124  KGVAE.forward
  /root/reference/kgvae/model.py:82-87    KGVAE.get_kl
  /root/reference/kgvae/model.py:71-80, :89-102  compute_kernel / get_mmd
  /root/reference/kgvae/model.py:60-69    KGVAE.sample_z
  /root/reference/kgvae/model.py:176-179, :203-211  RGCN.forward
  /root/reference/kgvae/link_predict.py:57-63   calc_score (DistMult)
  /root/reference/kgvae/link_predict.py:68-92   regularization_loss / get_loss

``state`` is a dict with the reference's LinkPredict ``state_dict`` keys
(``w_relation``, ``encoder.z_pre``, ``encoder.input_layer.embedding.weight``,
``encoder.rconv_layer_{1,2}.{weight,h_bias,loop_weight}``, ``encoder.nf.*``).
All randomness is an explicit input (``eps``, ``eps_prior``, ``post_idx``, dropout masks).
"""
import torch
import torch.nn.functional as F

from . import flows, prob, rgcn


def _layer_params(state, prefix):
    return {k: state[prefix + k] for k in ('weight', 'h_bias', 'loop_weight', 'w_comp')
            if prefix + k in state}


def kgvae_encode(state, src, dst, node_id, etypes, norm, eps, num_bases, n_flows=0,
                 dropout_p=0.0, keep1=None, keep2=None, prefix='encoder.'):
    """KGVAE.forward.  Returns dict(z, z_mean, z_sigma, flow_log_prob (0-d or None), h1, h2)."""
    ids = node_id.reshape(-1)
    x = state[prefix + 'input_layer.embedding.weight'].index_select(0, ids)
    h1 = rgcn.rel_graph_conv(x, src, dst, etypes, norm, _layer_params(state, prefix + 'rconv_layer_1.'),
                             'bdd', num_bases, activation=torch.relu,
                             dropout_keep=keep1, dropout_p=dropout_p)
    h2 = rgcn.rel_graph_conv(h1, src, dst, etypes, norm, _layer_params(state, prefix + 'rconv_layer_2.'),
                             'bdd', num_bases, activation=lambda t: t,
                             dropout_keep=keep2, dropout_p=dropout_p)
    z_mean, z_sigma = prob.gaussian_parameters(h2)
    z = prob.sample_gaussian(z_mean, z_sigma, eps)
    flp = None
    if n_flows > 0:
        z, log_det_sum = flows.flow_chain_forward(z, state, prefix + 'nf.', n_flows, z.shape[1])
        flp = torch.mean(log_det_sum.view(-1, 1))
    return dict(z=z, z_mean=z_mean, z_sigma=z_sigma, flow_log_prob=flp, h1=h1, h2=h2)


def rgcn_encode(state, src, dst, node_id, etypes, norm, num_bases, num_hidden_layers,
                dropout_p=0.0, keeps=None, prefix='encoder.'):
    """RGCN.forward: embedding + ``num_hidden_layers`` bdd layers, ReLU on all but the last."""
    h = state[prefix + 'layers.0.embedding.weight'].index_select(0, node_id.reshape(-1))
    for li in range(num_hidden_layers):
        act = torch.relu if li < num_hidden_layers - 1 else None
        keep = None if keeps is None else keeps[li]
        h = rgcn.rel_graph_conv(h, src, dst, etypes, norm, _layer_params(state, f'{prefix}layers.{li + 1}.'),
                                'bdd', num_bases, activation=act, dropout_keep=keep, dropout_p=dropout_p)
    return h


def kl_term(z, z_mean, z_sigma, z_pre, flow_log_prob):
    """get_kl.  ``flow_log_prob=None`` (the reference crashes there, SURVEY 0 bug 1) is read as 0."""
    m_mix, v_mix = prob.gaussian_parameters(z_pre, dim=1)
    flp = 0.0 if flow_log_prob is None else flow_log_prob
    return torch.mean(prob.log_normal(z, z_mean, z_sigma) + flp - prob.log_normal_mixture(z, m_mix, v_mix))


def rbf_kernel(x, y):
    """compute_kernel: exp(-mean_d (x-y)^2 / dim)."""
    dim = x.size(1)
    d2 = (x.unsqueeze(1) - y.unsqueeze(0)).pow(2).mean(2) / float(dim)
    return torch.exp(-d2)


def mmd_term(z, state, k, n_flows, eps_prior, post_idx, num_sample=200, prefix='encoder.'):
    """get_mmd.  ``eps_prior`` (num_sample//k*k, h) is the randn draw of sample_gaussian,
    ``post_idx`` the python ``random.sample(range(N), num_sample)`` pick."""
    m_mix, v_mix = prob.gaussian_parameters(state[prefix + 'z_pre'], dim=1)
    z_pri = prob.sample_gaussian(m_mix, v_mix, eps_prior, repeat=num_sample // k)
    if n_flows > 0:
        z_pri, _ = flows.flow_chain_forward(z_pri, state, prefix + 'nf.', n_flows, z_pri.shape[1])
    z_post = z[post_idx]
    return (rbf_kernel(z_pri, z_pri).mean() + rbf_kernel(z_post, z_post).mean()
            - 2 * rbf_kernel(z_pri, z_post).mean())


def sample_z(state, comp_idx, eps, n_flows, prefix='encoder.'):
    """KGVAE.sample_z with the Categorical draw ``comp_idx`` and randn draw ``eps`` given."""
    m, v = prob.gaussian_parameters(state[prefix + 'z_pre'].squeeze(0), dim=0)
    x = prob.sample_gaussian(m[comp_idx], v[comp_idx], eps)
    if n_flows > 0:
        x = flows.flow_chain_inverse(x, state, prefix + 'nf.', n_flows, x.shape[1])
    return x


def distmult_score(embedding, w_relation, triplets):
    s = embedding[triplets[:, 0]]
    r = w_relation[triplets[:, 1]]
    o = embedding[triplets[:, 2]]
    return torch.sum(s * r * o, dim=1)


def link_predict_loss(state, enc, triplets, labels, reg_param, kl_param, mmd_param, k, n_flows,
                      eps_prior=None, post_idx=None):
    """LinkPredict.get_loss for a KGVAE encoder.  Returns (loss, predict_loss, kl, mmd)."""
    embed = enc['z']
    score = distmult_score(embed, state['w_relation'], triplets)
    if n_flows > 0:
        score = score + enc['flow_log_prob']
    predict_loss = F.binary_cross_entropy_with_logits(score, labels)
    reg = torch.mean(embed.pow(2)) + torch.mean(state['w_relation'].pow(2))
    if kl_param > 0:
        kl = kl_term(embed, enc['z_mean'], enc['z_sigma'], state['encoder.z_pre'], enc['flow_log_prob'])
    else:
        kl = torch.zeros(1)
    if mmd_param > 0:
        mmd = mmd_term(embed, state, k, n_flows, eps_prior, post_idx)
    else:
        mmd = torch.zeros(1)
    loss = predict_loss + reg_param * reg + kl_param * kl + mmd_param * mmd
    return loss, predict_loss, kl, mmd


def rgcn_link_predict_loss(state, embed, triplets, labels, reg_param):
    """get_loss for the plain RGCN encoder (kl = mmd = 0; the reference's
    ``BaseRGCN.get_kl`` returns a zero and ``LinkPredict`` crashes before reaching it,
    SURVEY 0 bug 2)."""
    score = distmult_score(embed, state['w_relation'], triplets)
    predict_loss = F.binary_cross_entropy_with_logits(score, labels)
    reg = torch.mean(embed.pow(2)) + torch.mean(state['w_relation'].pow(2))
    return predict_loss + reg_param * reg, predict_loss
