"""``RelGraphConv`` -- the R-GCN layer of DGL 0.4.x, re-implemented on the gfx950 K1/K2 kernels.

Interface and state_dict keys follow the DGL layer as the reference uses it
(kgvae/model.py:54-59, :110-111, :209-211; kgvae/entity_classify.py:31-43):

    RelGraphConv(in_feat, out_feat, num_rels, regularizer="basis", num_bases=None, bias=True,
                 activation=None, self_loop=False, dropout=0.0).forward(g, x, etypes, norm=None)

parameters ``weight`` [(R, B*si*so) for "bdd", (nb, in, out) for "basis"], ``w_comp`` (basis, nb < R),
``h_bias`` (zeros), ``loop_weight``; xavier_uniform with relu gain, created in DGL's order so that a
seeded construction reproduces the reference's initial weights.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import graph_index_of


def _activation_id(activation):
    """Map the reference's activations onto the fused epilogue; anything else runs after the kernel."""
    if activation is None:
        return ops.ACT_NONE, None
    if activation is F.relu or activation is torch.relu or isinstance(activation, nn.ReLU):
        return ops.ACT_RELU, None
    return ops.ACT_NONE, activation


class RelGraphConv(ops.StayOnDevice, nn.Module):
    def __init__(self, in_feat, out_feat, num_rels, regularizer="basis", num_bases=None, bias=True,
                 activation=None, self_loop=False, dropout=0.0):
        super().__init__()
        self.in_feat, self.out_feat, self.num_rels = in_feat, out_feat, num_rels
        self.regularizer = regularizer
        self.num_bases = num_bases
        if self.num_bases is None or self.num_bases > self.num_rels or self.num_bases < 0:
            self.num_bases = self.num_rels
        self.bias, self.activation, self.self_loop = bias, activation, self_loop
        gain = nn.init.calculate_gain('relu')
        if regularizer == "basis":
            self.weight = nn.Parameter(torch.Tensor(self.num_bases, in_feat, out_feat))
            if self.num_bases < self.num_rels:
                self.w_comp = nn.Parameter(torch.Tensor(self.num_rels, self.num_bases))
            nn.init.xavier_uniform_(self.weight, gain=gain)
            if self.num_bases < self.num_rels:
                nn.init.xavier_uniform_(self.w_comp, gain=gain)
        elif regularizer == "bdd":
            if in_feat % self.num_bases != 0 or out_feat % self.num_bases != 0:
                raise ValueError('Feature size must be a multiplier of num_bases.')
            self.submat_in = in_feat // self.num_bases
            self.submat_out = out_feat // self.num_bases
            self.weight = nn.Parameter(torch.Tensor(self.num_rels, self.num_bases * self.submat_in * self.submat_out))
            nn.init.xavier_uniform_(self.weight, gain=gain)
        else:
            raise ValueError("Regularizer must be either 'basis' or 'bdd'")
        if self.bias:
            self.h_bias = nn.Parameter(torch.Tensor(out_feat))
            nn.init.zeros_(self.h_bias)
        if self.self_loop:
            self.loop_weight = nn.Parameter(torch.Tensor(in_feat, out_feat))
            nn.init.xavier_uniform_(self.loop_weight, gain=gain)
        self.dropout = nn.Dropout(dropout)
        self.keep_mask_override = None   # parity mode: a uint8 (N, out) keep mask instead of the device RNG
        self.rng_stream = ops.new_rng_stream()
        self._keep_next = None           # a mask the enclosing encoder drew for this call (one fused RNG launch)
        self.reduce_hook = None          # multi-GPU: sums the partial aggregate over the edge shards

    def _keep_mask(self, n, device):
        p = self.dropout.p
        if self.keep_mask_override is not None:
            return self.keep_mask_override.to(device=device, dtype=torch.uint8).contiguous(), 1.0 / (1.0 - p)
        if not self.training or p <= 0.0:
            return None, 1.0
        if self._keep_next is not None:
            keep, self._keep_next = self._keep_next, None
            return keep, 1.0 / (1.0 - p)
        keep = torch.empty(n, self.out_feat, dtype=torch.uint8, device=device)
        ops.device_rng(device).fill([self.keep_job(keep)])
        return keep, 1.0 / (1.0 - p)

    def wants_keep_mask(self):
        return self.training and self.dropout.p > 0.0 and self.keep_mask_override is None

    def keep_job(self, keep):
        """The RNG job that fills ``keep`` (uint8, (N, out_feat)) with this layer's Bernoulli(1 - p) decisions."""
        return (keep, ops.RNG_KEEP_MASK, float(self.dropout.p), self.rng_stream)

    def forward_rows(self, g, x, etypes, norm, part, gather_input, pad_output, gather_output=False, x_gathered=False):
        """The layer on ONE rank's row block of the multi-GPU destination-row partition (ops.rel_graph_conv_rows):
        ``g`` is the rank's distributed.RowBlockGraph, ``x`` the full table (gather_input False) or the rank's slot.
        gather_output / x_gathered: the pipelined exchange between two layers (the producer gathers its rows block by block
        under its own aggregation and returns the full table; the consumer reduce-scatters its backward the same way)."""
        if self.regularizer != 'bdd':
            raise NotImplementedError('the row partition covers the bdd regulariser (the reference encoders use only it)')
        gidx = graph_index_of(g, x.device)
        ridx = gidx.relation_index(etypes, self.num_rels)
        act_id, post_act = _activation_id(self.activation if self.activation else None)
        if post_act is not None:
            raise NotImplementedError('row partition: activation must be None or ReLU')
        c = part.own_rows
        keep, scale = self._keep_mask(c, x.device)
        if keep is not None and keep.shape[0] != c:       # parity override given for all positions
            keep = keep[part.row0:part.row0 + c].contiguous()
        return ops.rel_graph_conv_rows(x, self.weight, self.h_bias if self.bias else None,
                                       self.loop_weight if self.self_loop else None, norm, gidx, ridx, self.num_bases,
                                       part, act_id, keep, scale if keep is not None else 1.0, gather_input, pad_output,
                                       gather_output, x_gathered)

    def forward(self, g, x, etypes, norm=None):
        x, etypes, norm = ops.to_module_device(self.weight, x, etypes, norm)
        int_ids = x.dtype == torch.int64 and x.dim() == 1
        if int_ids and self.regularizer == 'bdd':
            raise TypeError('Block decomposition does not allow integer ID feature.')
        gidx = graph_index_of(g, x.device)
        ridx = gidx.relation_index(etypes, self.num_rels)
        act_id, post_act = _activation_id(self.activation if self.activation else None)
        keep, scale = self._keep_mask(x.shape[0], x.device)
        h_bias = self.h_bias if self.bias else None
        loop_w = self.loop_weight if self.self_loop else None
        if post_act is not None and keep is not None:      # unknown activation: dropout must follow it
            late_keep, keep = keep, None
        else:
            late_keep = None
        if int_ids:
            # integer-id features (kgvae/entity_classify.py:25-34, :63: features = arange(num_nodes) into a basis layer
            # with in_feat = num_nodes): a message is a ROW of the relation's matrix (DGL bmm_maybe_select), the
            # self-loop term a row of loop_weight (matmul_maybe_select)
            if self.reduce_hook is not None:
                raise NotImplementedError('integer-id features are not wired into the multi-GPU edge sharding')
            flat = self.weight.view(self.num_bases, self.in_feat * self.out_feat)
            weight = ops.matmul(self.w_comp, flat) if self.num_bases < self.num_rels else flat
            h = ops.rel_graph_conv_select(x, weight.view(self.num_rels, self.in_feat, self.out_feat), h_bias, loop_w, norm,
                                          gidx, ridx, act_id, keep, scale if keep is not None else 1.0)
        elif self.regularizer == 'bdd':
            h = ops.rel_graph_conv_bdd(x, self.weight, h_bias, loop_w, norm, gidx, ridx, self.num_bases, act_id, keep,
                                       scale if keep is not None else 1.0, self.reduce_hook)
        else:
            # basis: W_r = sum_b w_comp[r, b] V_b (one MFMA GEMM), then a full (in x out) matrix per relation
            flat = self.weight.view(self.num_bases, self.in_feat * self.out_feat)
            weight = ops.matmul(self.w_comp, flat) if self.num_bases < self.num_rels else flat
            if self.reduce_hook is not None or os.environ.get('GV_BASIS_GENERIC', '0') == '1':
                # edge-sharded multi-GPU hook / cross-check: the generic K1 kernels (one dense "block" per relation)
                h = ops.rel_graph_conv_bdd(x, weight, h_bias, loop_w, norm, gidx, ridx, 1, act_id, keep,
                                           scale if keep is not None else 1.0, self.reduce_hook)
            else:
                h = ops.rel_graph_conv_dense(x, weight.view(self.num_rels, self.in_feat, self.out_feat), h_bias, loop_w, norm,
                                             gidx, ridx, act_id, keep, scale if keep is not None else 1.0)
        if post_act is not None:
            h = post_act(h)
            if late_keep is not None:
                h = h * (late_keep.to(h.dtype) * scale)
        return h
