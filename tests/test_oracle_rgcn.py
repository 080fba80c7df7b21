"""Independent cross-checks of the (parity-unpinned) RelGraphConv restatement  --  SURVEY.md 8(c).

(1) dense per-relation adjacency formulation, (2) scalar loops, (3) fp64 gradcheck,
(4) properties: edge-permutation invariance, zero in-degree rows, B=1, basis with identity w_comp."""
import numpy as np
import pytest
import torch

from oracle import rgcn


def make_case(n, e, r, fin, fout, nb, seed, reg='bdd', dtype=torch.float32, self_loop=True):
    gen = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, max(1, n - 2), (e,), generator=gen)       # last nodes get no in-edges
    et = torch.randint(0, r, (e,), generator=gen)
    deg = torch.bincount(dst, minlength=n).clamp(min=1).to(dtype)
    norm = (1.0 / deg)[dst].view(-1, 1)
    x = torch.randn(n, fin, generator=gen, dtype=dtype)
    p = rgcn.init_params(fin, fout, r, reg, nb, True, self_loop, gen, dtype)
    p['h_bias'] = torch.randn(fout, generator=gen, dtype=dtype) * 0.1
    return x, src, dst, et, norm, p


@pytest.mark.parametrize('fin,fout,nb', [(8, 8, 4), (8, 16, 4), (6, 6, 6), (10, 20, 2), (4, 4, 1)])
def test_bdd_three_formulations_agree(fin, fout, nb):
    x, src, dst, et, norm, p = make_case(17, 60, 7, fin, fout, nb, seed=fin * 100 + nb)
    a = rgcn.rel_graph_conv(x, src, dst, et, norm, p, 'bdd', nb, torch.relu)
    b = rgcn.rel_graph_conv_dense(x, src, dst, et, norm, p, 'bdd', nb, torch.relu)
    c = rgcn.rel_graph_conv_loops(x, src, dst, et, norm, p, nb, torch.relu)
    torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-5)


def test_basis_matches_dense_and_identity_comp():
    x, src, dst, et, norm, p = make_case(15, 50, 6, 5, 7, 3, seed=3, reg='basis')
    assert p['weight'].shape == (3, 5, 7) and p['w_comp'].shape == (6, 3)
    a = rgcn.rel_graph_conv(x, src, dst, et, norm, p, 'basis', 3, torch.tanh)
    b = rgcn.rel_graph_conv_dense(x, src, dst, et, norm, p, 'basis', 3, torch.tanh)
    torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    # nb == R: no w_comp, weight is used per relation directly
    x, src, dst, et, norm, p = make_case(15, 50, 6, 5, 7, 6, seed=4, reg='basis')
    assert 'w_comp' not in p
    q = dict(p, w_comp=torch.eye(6))
    torch.testing.assert_close(rgcn.rel_graph_conv(x, src, dst, et, norm, p, 'basis', 6),
                               rgcn.rel_graph_conv(x, src, dst, et, norm, q, 'basis', 6), rtol=1e-6, atol=1e-6)


def test_num_bases_clamp_and_divisibility_error():
    assert rgcn.clamp_num_bases(None, 7) == 7 and rgcn.clamp_num_bases(-1, 7) == 7
    assert rgcn.clamp_num_bases(100, 22) == 22 and rgcn.clamp_num_bases(4, 7) == 4
    with pytest.raises(ValueError, match='multiplier of num_bases'):
        rgcn.init_params(200, 200, 22, 'bdd', 100)          # C3: clamped to 22, 200 % 22 != 0
    with pytest.raises(ValueError):
        rgcn.init_params(8, 8, 3, 'nope', 2)


def test_properties():
    x, src, dst, et, norm, p = make_case(20, 80, 4, 8, 8, 4, seed=9)
    base = rgcn.rel_graph_conv(x, src, dst, et, norm, p, 'bdd', 4, torch.relu)
    perm = torch.randperm(80, generator=torch.Generator().manual_seed(1))
    shuf = rgcn.rel_graph_conv(x, src[perm], dst[perm], et[perm], norm[perm], p, 'bdd', 4, torch.relu)
    torch.testing.assert_close(base, shuf, rtol=1e-5, atol=1e-6)
    iso = torch.bincount(dst, minlength=20) == 0
    assert iso.any()
    torch.testing.assert_close(base[iso], torch.relu(p['h_bias'] + x[iso] @ p['loop_weight']), rtol=1e-6, atol=1e-6)
    # dropout mask semantics
    keep = (torch.rand(20, 8, generator=torch.Generator().manual_seed(2)) > 0.3).float()
    dr = rgcn.rel_graph_conv(x, src, dst, et, norm, p, 'bdd', 4, torch.relu, dropout_keep=keep, dropout_p=0.3)
    torch.testing.assert_close(dr, base * keep / 0.7, rtol=1e-6, atol=1e-6)


def test_gradcheck_fp64():
    x, src, dst, et, norm, p = make_case(7, 20, 3, 4, 6, 2, seed=5, dtype=torch.float64)
    x.requires_grad_(True)
    w, b, lw = (p[k].clone().requires_grad_(True) for k in ('weight', 'h_bias', 'loop_weight'))

    def f(x_, w_, b_, lw_):
        return rgcn.rel_graph_conv(x_, src, dst, et, norm, dict(weight=w_, h_bias=b_, loop_weight=lw_), 'bdd', 2,
                                   torch.tanh)
    assert torch.autograd.gradcheck(f, (x, w, b, lw), eps=1e-6, atol=1e-5)


def test_integer_id_features_equal_one_hot_features():
    """DGL bmm_maybe_select / matmul_maybe_select (kgvae/entity_classify.py:25-34, :63): integer ids select rows -- the same
    numbers as multiplying the one-hot matrix of those ids, forward and gradients."""
    gen = torch.Generator().manual_seed(0)
    n, e, r, fout, nb = 40, 300, 6, 8, 3
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, n, (e,), generator=gen)
    et = torch.randint(0, r, (e,), generator=gen)
    norm = torch.rand(e, 1, generator=gen)
    ids = torch.randperm(n, generator=gen)
    p = rgcn.init_params(n, fout, r, 'basis', nb, True, True, gen)
    p['h_bias'] = torch.randn(fout, generator=gen) * 0.1
    pa = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    pb = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ha = rgcn.rel_graph_conv(ids, src, dst, et, norm, pa, 'basis', nb, torch.relu)
    hb = rgcn.rel_graph_conv(torch.eye(n)[ids], src, dst, et, norm, pb, 'basis', nb, torch.relu)
    gout = torch.randn(n, fout, generator=gen)
    ha.backward(gout)
    hb.backward(gout)
    torch.testing.assert_close(ha, hb, rtol=1e-5, atol=1e-6)
    for k in pa:
        torch.testing.assert_close(pa[k].grad, pb[k].grad, rtol=1e-4, atol=1e-6)
    with pytest.raises(TypeError):
        rgcn.rel_graph_conv(ids, src, dst, et, norm, rgcn.init_params(8, 8, r, 'bdd', 2, True, True, gen), 'bdd', 2)


def test_chunked_aggregate_equals_the_materialised_op_sequence(monkeypatch):
    """The edge-chunked evaluation the oracle switches to when the per-edge weight gather would not fit (emb_dim = 500 at
    FB15k-237 size) is the same function: bit-identical forward, gradients equal to rounding."""
    from oracle import rgcn as orgcn
    gen = torch.Generator().manual_seed(3)
    n, e, r, fin, fout, nb = 50, 700, 6, 20, 40, 4
    src, dst, et = (torch.randint(0, n, (e,), generator=gen), torch.randint(0, n, (e,), generator=gen),
                    torch.randint(0, r, (e,), generator=gen))
    norm = torch.rand(e, 1, generator=gen)
    p = orgcn.init_params(fin, fout, r, 'bdd', nb, True, True, gen)
    x = torch.randn(n, fin, generator=gen)
    gout = torch.randn(n, fout, generator=gen)
    outs = []
    for limit, chunk in ((1 << 40, 40000), (0, 64)):
        monkeypatch.setattr(orgcn, 'MATERIALISE_LIMIT_BYTES', limit)
        monkeypatch.setattr(orgcn, 'EDGE_CHUNK', chunk)
        xo = x.clone().requires_grad_(True)
        po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        h = orgcn.rel_graph_conv(xo, src, dst, et, norm, po, 'bdd', nb, torch.relu)
        h.backward(gout)
        outs.append((h.detach(), xo.grad, po['weight'].grad, po['loop_weight'].grad))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1:], outs[1][1:]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
