"""Oracle restatement of ``dgl.nn.pytorch.RelGraphConv`` (DGL 0.4.x)  --  PARITY UNPINNED.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  The layer's source is not in
``/root/reference``; the reference only calls it:

* ctor call sites   kgvae/model.py:54-56 (h->h, ReLU, self_loop), :57-59 (h->2h, identity),
                    kgvae/model.py:209-211 (RGCN), kgvae/entity_classify.py:31-43 (basis)
* forward call sites kgvae/model.py:110-111, :177-178

Semantics restated from DGL 0.4.x ``python/dgl/nn/pytorch/conv/relgraphconv.py``:

  bdd   : W_e = weight.index_select(0, etype).view(-1, si, so)
          msg = bmm(x[src].view(-1, 1, si), W_e).view(-1, out) * norm
  basis : W   = (w_comp @ weight.view(nb, in*out)).view(R, in, out)   (if nb < R)
          msg = bmm(x[src].unsqueeze(1), W[etype]).squeeze(1) * norm
          integer-id features (x int64 (N,), the one-hot input layer of kgvae/entity_classify.py:25, :31-34, :63 --
          ``features = torch.arange(num_nodes)`` into ``RelGraphConv(num_nodes, h, R, "basis", ...)``):
          msg = W.view(-1, out)[etype * in + x[src]] * norm     (DGL utils.bmm_maybe_select: a row select, no product)
          and the self-loop term is loop_weight[x]               (DGL utils.matmul_maybe_select)
  h[v]  = sum of msg over in-edges of v (0 for in-degree 0)
  h     = h + h_bias ; h = h + x @ loop_weight ; h = activation(h) ; h = dropout(h)

Three implementations live here so they can be checked against each other:
``rel_graph_conv`` (the reference op sequence, also the CPU timing baseline),
``rel_graph_conv_dense`` (explicit per-relation adjacency matrices) and
``rel_graph_conv_loops`` (scalar python loops, tiny inputs only).
"""
import torch

from . import bf16


def clamp_num_bases(num_bases, num_rels):
    """DGL 0.4.x: ``None``, negative or > num_rels  ==>  num_rels."""
    if num_bases is None or num_bases > num_rels or num_bases < 0:
        return num_rels
    return num_bases


def bdd_block_sizes(in_feat, out_feat, num_bases):
    if in_feat % num_bases != 0 or out_feat % num_bases != 0:
        raise ValueError('Feature size must be a multiplier of num_bases.')
    return in_feat // num_bases, out_feat // num_bases


def init_params(in_feat, out_feat, num_rels, regularizer, num_bases, bias=True,
                self_loop=False, generator=None, dtype=torch.float32):
    """Parameter shapes + initialisers of the DGL layer (xavier_uniform, gain=relu; zero bias)."""
    gain = torch.nn.init.calculate_gain('relu')
    nb = clamp_num_bases(num_bases, num_rels)
    p = {}
    if regularizer == 'bdd':
        si, so = bdd_block_sizes(in_feat, out_feat, nb)
        p['weight'] = torch.empty(num_rels, nb * si * so, dtype=dtype)
        torch.nn.init.xavier_uniform_(p['weight'], gain=gain, generator=generator)
    elif regularizer == 'basis':
        p['weight'] = torch.empty(nb, in_feat, out_feat, dtype=dtype)
        torch.nn.init.xavier_uniform_(p['weight'], gain=gain, generator=generator)
        if nb < num_rels:
            p['w_comp'] = torch.empty(num_rels, nb, dtype=dtype)
            torch.nn.init.xavier_uniform_(p['w_comp'], gain=gain, generator=generator)
    else:
        raise ValueError("Regularizer must be either 'basis' or 'bdd'")
    if bias:
        p['h_bias'] = torch.zeros(out_feat, dtype=dtype)
    if self_loop:
        p['loop_weight'] = torch.empty(in_feat, out_feat, dtype=dtype)
        torch.nn.init.xavier_uniform_(p['loop_weight'], gain=gain, generator=generator)
    return p


def _messages(x, src, etypes, norm, params, regularizer, num_bases):
    num_rels = params['w_comp'].shape[0] if 'w_comp' in params else params['weight'].shape[0]
    if regularizer == 'bdd':
        w = params['weight']
        nb = clamp_num_bases(num_bases, w.shape[0])
        in_feat = x.shape[1]
        si = in_feat // nb
        so = w.shape[1] // (nb * si)
        w_e = w.index_select(0, etypes).view(-1, si, so)          # (E*B, si, so)  materialised
        node = x.index_select(0, src).view(-1, 1, si)             # (E*B, 1, si)
        msg = torch.bmm(node, w_e).view(-1, nb * so)
    elif regularizer == 'basis':
        w = params['weight']
        if 'w_comp' in params:
            nbv, fi, fo = w.shape
            w = torch.matmul(params['w_comp'], w.view(nbv, fi * fo)).view(num_rels, fi, fo)
        if x.dtype == torch.int64 and x.dim() == 1:       # integer ids: row (etype, id) of the relation's matrix
            msg = w.reshape(-1, w.shape[2]).index_select(0, etypes * w.shape[1] + x.index_select(0, src))
        else:
            msg = torch.bmm(x.index_select(0, src).unsqueeze(1), w.index_select(0, etypes)).squeeze(1)
    else:
        raise ValueError("Regularizer must be either 'basis' or 'bdd'")
    if norm is not None:
        msg = msg * norm.view(-1, 1)
    return msg


MATERIALISE_LIMIT_BYTES = 3 << 30      # above this the per-edge weight gather of a bdd layer is evaluated over edge chunks
EDGE_CHUNK = 40000


class _ChunkedBddAggregate(torch.autograd.Function):
    """``index_add(dst, bmm(x[src], W[etypes]) * norm)`` -- the reference's op sequence -- WITHOUT its E x (in*out/B) weight
    gather alive at once (5.4 / 10.9 GB at emb_dim = 500 on FB15k-237, several times that under autograd): the aggregate is
    a sum over edges, so it is accumulated chunk by chunk in the same edge order (bit-identical forward); backward
    recomputes each chunk's gather.  Same arithmetic, bounded memory."""

    @staticmethod
    def forward(ctx, x, w, src, dst, etypes, norm, nb):
        si = x.shape[1] // nb
        so = w.shape[1] // (nb * si)
        agg = torch.zeros(x.shape[0], nb * so, dtype=x.dtype)
        for c0 in range(0, src.numel(), EDGE_CHUNK):
            sl = slice(c0, c0 + EDGE_CHUNK)
            msg = torch.bmm(x.index_select(0, src[sl]).view(-1, 1, si), w.index_select(0, etypes[sl]).view(-1, si, so)).view(-1, nb * so)
            if norm is not None:
                msg = msg * norm[sl].view(-1, 1)
            agg.index_add_(0, dst[sl], msg)
        ctx.save_for_backward(x, w, src, dst, etypes, norm if norm is not None else torch.zeros(0))
        ctx.nb, ctx.has_norm = nb, norm is not None
        return agg

    @staticmethod
    def backward(ctx, g):
        x, w, src, dst, etypes, norm = ctx.saved_tensors
        nb = ctx.nb
        si = x.shape[1] // nb
        so = w.shape[1] // (nb * si)
        gx, gw = torch.zeros_like(x), torch.zeros_like(w)
        for c0 in range(0, src.numel(), EDGE_CHUNK):
            sl = slice(c0, c0 + EDGE_CHUNK)
            ge = g.index_select(0, dst[sl])
            if ctx.has_norm:
                ge = ge * norm[sl].view(-1, 1)
            ge = ge.view(-1, 1, so)
            xe = x.index_select(0, src[sl]).view(-1, si, 1)
            gx.index_add_(0, src[sl], torch.bmm(ge, w.index_select(0, etypes[sl]).view(-1, si, so).transpose(1, 2)).view(-1, nb * si))
            gw.index_add_(0, etypes[sl], torch.bmm(xe, ge).view(-1, nb * si * so))
        return gx, gw, None, None, None, None, None


def rel_graph_conv(x, src, dst, etypes, norm, params, regularizer='bdd', num_bases=None,
                   activation=None, dropout_keep=None, dropout_p=0.0):
    """Reference op sequence.  ``src, dst, etypes`` int64 (E,), ``norm`` (E,1)/(E,) or None.

    ``dropout_keep``: optional 0/1 tensor (N,out); the output is multiplied by
    ``keep / (1 - dropout_p)`` (what ``nn.Dropout`` does in training mode with that mask).
    """
    int_ids = x.dtype == torch.int64 and x.dim() == 1
    if int_ids and regularizer == 'bdd':
        raise TypeError('Block decomposition does not allow integer ID feature.')
    if regularizer == 'bdd' and bf16.k1_enabled():      # the product's bf16-operand aggregation (oracle/bf16.py)
        h = bf16.bdd_aggregate(x, params['weight'], src, dst, etypes, norm, clamp_num_bases(num_bases, params['weight'].shape[0]))
    elif regularizer == 'bdd' and src.numel() * params['weight'].shape[1] * 4 > MATERIALISE_LIMIT_BYTES:
        h = _ChunkedBddAggregate.apply(x, params['weight'], src, dst, etypes, norm,
                                       clamp_num_bases(num_bases, params['weight'].shape[0]))
    else:
        msg = _messages(x, src, etypes, norm, params, regularizer, num_bases)
        h = torch.zeros(x.shape[0], msg.shape[1], dtype=msg.dtype).index_add(0, dst, msg)
    if 'h_bias' in params:
        h = h + params['h_bias']
    if 'loop_weight' in params:
        if int_ids:
            h = h + params['loop_weight'].index_select(0, x)
        else:
            h = h + bf16.mm(x, params['loop_weight'])      # fp32 unless oracle.bf16.enabled()
    if activation is not None:
        h = activation(h)
    if dropout_keep is not None:
        h = h * dropout_keep.to(h.dtype) / (1.0 - dropout_p)
    return h


def full_relation_weights(params, regularizer, num_bases, in_feat):
    """(R, in, out) dense per-relation matrices (block-diagonal expanded for bdd)."""
    w = params['weight']
    if regularizer == 'bdd':
        num_rels = w.shape[0]
        nb = clamp_num_bases(num_bases, num_rels)
        si = in_feat // nb
        so = w.shape[1] // (nb * si)
        blocks = w.view(num_rels, nb, si, so)
        full = torch.zeros(num_rels, nb * si, nb * so, dtype=w.dtype)
        for b in range(nb):
            full[:, b * si:(b + 1) * si, b * so:(b + 1) * so] = blocks[:, b]
        return full
    if 'w_comp' in params:
        nbv, fi, fo = w.shape
        return torch.einsum('rb,bio->rio', params['w_comp'], w)
    return w


def rel_graph_conv_dense(x, src, dst, etypes, norm, params, regularizer='bdd', num_bases=None,
                         activation=None):
    """Independent formulation:  out = sum_r  A_r @ X @ W_r  + b + X W_loop,
    A_r[v,u] = sum of norm over edges u->v of type r."""
    n = x.shape[0]
    wfull = full_relation_weights(params, regularizer, num_bases, x.shape[1])
    num_rels = wfull.shape[0]
    nrm = torch.ones(src.shape[0], dtype=x.dtype) if norm is None else norm.view(-1).to(x.dtype)
    h = torch.zeros(n, wfull.shape[2], dtype=x.dtype)
    for r in range(num_rels):
        sel = etypes == r
        if not bool(sel.any()):
            continue
        a = torch.zeros(n, n, dtype=x.dtype)
        a.index_put_((dst[sel], src[sel]), nrm[sel], accumulate=True)
        h = h + a @ (x @ wfull[r])
    if 'h_bias' in params:
        h = h + params['h_bias']
    if 'loop_weight' in params:
        h = h + x @ params['loop_weight']
    return activation(h) if activation is not None else h


def rel_graph_conv_loops(x, src, dst, etypes, norm, params, num_bases=None, activation=None):
    """bdd only, scalar loops (tiny inputs)."""
    w = params['weight']
    num_rels = w.shape[0]
    nb = clamp_num_bases(num_bases, num_rels)
    n, fin = x.shape
    si = fin // nb
    so = w.shape[1] // (nb * si)
    out = [[0.0] * (nb * so) for _ in range(n)]
    for e in range(src.shape[0]):
        u, v, r = int(src[e]), int(dst[e]), int(etypes[e])
        c = 1.0 if norm is None else float(norm.view(-1)[e])
        for b in range(nb):
            for j in range(so):
                acc = 0.0
                for i in range(si):
                    acc += float(x[u, b * si + i]) * float(w[r, (b * si + i) * so + j])
                out[v][b * so + j] += c * acc
    h = torch.tensor(out, dtype=x.dtype)
    if 'h_bias' in params:
        h = h + params['h_bias']
    if 'loop_weight' in params:
        h = h + x @ params['loop_weight']
    return activation(h) if activation is not None else h
