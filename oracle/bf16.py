"""TEST INFRASTRUCTURE ONLY -- emulation of the product's bf16-operand GEMMs (include/gcnvae.h: gv_gemm_bf16) for the
CPU oracle.  PARITY UNPINNED: the reference has no reduced-precision path (BASELINE configs[2] names bf16 as the
target precision of the port, kgvae/* is fp32 throughout), so this file DEFINES the semantics the HIP path is held to:

    forward   y  = r(a) @ r(b)                 r = round-to-nearest-even to bfloat16, product and sum in fp32
    backward  ga = r(g) @ r(b)^T ,  gb = r(a)^T @ r(g)

i.e. every dense product of the step -- MaskedLinear (kgvae/flow_network.py:14-15) and the self-loop term of
RelGraphConv -- takes bf16 operands and accumulates in fp32; everything else stays fp32.  Disabled (the default) the
helpers are exactly the fp32 calls the oracle made before, so the golden fixtures are untouched.

``enabled(on, k1=True)`` additionally puts the R-GCN AGGREGATION on bf16 operands, as the product's LDS-resident K1 kernel
does for graphs with few relation types (include/gcnvae.h: gv_rgcn_bdd_aggregate_lds, bf16_operands = 1):

    forward     agg[v] = sum_{e: dst = v}  r(norm_e x[src_e]) . blockdiag(r(W_{type_e}))        products and sums in fp32
    backward-x  gx[u]  = sum_{e: src = u}  r(norm_e g[dst_e]) . blockdiag(r(W_{type_e}))^T
    backward-W  fp32 throughout (the weight-gradient kernel is not touched by the switch)
"""
import contextlib

import torch
import torch.nn.functional as F

_enabled = False
_k1 = False


def _r(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _MM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ra, rb = _r(a), _r(b)
        ctx.save_for_backward(ra, rb)
        return ra @ rb

    @staticmethod
    def backward(ctx, g):
        ra, rb = ctx.saved_tensors
        rg = _r(g)
        return rg @ rb.t(), ra.t() @ rg


def mm(a, b):
    return _MM.apply(a, b) if _enabled else a @ b


def linear(x, w, b=None):
    if not _enabled:
        return F.linear(x, w, b)
    y = _MM.apply(x, w.t())
    return y if b is None else y + b


class _BddAggregate(torch.autograd.Function):
    """The block-diagonal aggregation on bf16 operands (see the module docstring)."""

    @staticmethod
    def forward(ctx, x, w, src, dst, etypes, norm, nb):
        n, fin = x.shape
        si = fin // nb
        so = w.shape[1] // (nb * si)
        c = torch.ones(src.numel(), 1) if norm is None else norm.reshape(-1, 1)
        wr = _r(w).index_select(0, etypes).view(-1, si, so)
        node = _r(x.index_select(0, src) * c).view(-1, 1, si)
        msg = torch.bmm(node, wr).view(-1, nb * so)
        ctx.save_for_backward(x, w, src, dst, etypes, c)
        ctx.nb = nb
        return torch.zeros(n, nb * so, dtype=x.dtype).index_add(0, dst, msg)

    @staticmethod
    def backward(ctx, g):
        x, w, src, dst, etypes, c = ctx.saved_tensors
        nb = ctx.nb
        n, fin = x.shape
        si = fin // nb
        so = w.shape[1] // (nb * si)
        ge = g.index_select(0, dst)
        wr = _r(w).index_select(0, etypes).view(-1, si, so)
        gx_e = torch.bmm(_r(ge * c).view(-1, 1, so), wr.transpose(1, 2)).view(-1, fin)
        gx = torch.zeros(n, fin, dtype=x.dtype).index_add(0, src, gx_e)
        del wr, gx_e
        gw_e = torch.bmm((x.index_select(0, src) * c).view(-1, si, 1), ge.view(-1, 1, so)).view(-1, nb * si * so)
        gw = torch.zeros_like(w).index_add(0, etypes, gw_e)
        return gx, gw, None, None, None, None, None


def k1_enabled():
    return _enabled and _k1


def bdd_aggregate(x, w, src, dst, etypes, norm, nb):
    return _BddAggregate.apply(x, w, src, dst, etypes, norm, nb)


@contextlib.contextmanager
def enabled(on=True, k1=False):
    global _enabled, _k1
    old, _enabled, _k1 = (_enabled, _k1), bool(on), bool(on) and bool(k1)
    try:
        yield
    finally:
        _enabled, _k1 = old
