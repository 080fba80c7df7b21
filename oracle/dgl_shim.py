"""Stand-in ``dgl`` package used ONLY to import the reference for fixture generation.

TEST INFRASTRUCTURE.  ``install()`` registers ``dgl``, ``dgl.nn``, ``dgl.nn.pytorch``,
``dgl.contrib``, ``dgl.contrib.data`` in ``sys.modules`` so that
``/root/reference/kgvae/{utils,model,link_predict}.py`` import in the build container.
``RelGraphConv`` here is an ``nn.Module`` face over ``oracle.rgcn`` with DGL's parameter
names, so whatever the reference computes *around* the layer is the reference's own code.
"""
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import graphs, rgcn


class RelGraphConv(nn.Module):
    def __init__(self, in_feat, out_feat, num_rels, regularizer="basis", num_bases=None, bias=True,
                 activation=None, self_loop=False, dropout=0.0):
        super().__init__()
        self.regularizer = regularizer
        self.num_bases = rgcn.clamp_num_bases(num_bases, num_rels)
        self.activation = activation
        for k, v in rgcn.init_params(in_feat, out_feat, num_rels, regularizer, num_bases, bias,
                                     self_loop).items():
            setattr(self, k, nn.Parameter(v))
        self.dropout = nn.Dropout(dropout)

    def forward(self, g, x, etypes, norm=None):
        src, dst = g.edges()
        params = {k: getattr(self, k) for k in ('weight', 'w_comp', 'h_bias', 'loop_weight')
                  if hasattr(self, k)}
        h = rgcn.rel_graph_conv(x, src, dst, etypes, norm, params, self.regularizer, self.num_bases,
                                activation=self.activation if self.activation else None)
        return self.dropout(h)


class SyntheticKG:
    """Object with the attributes ``load_data`` returns (kgvae/link_predict.py:105-110)."""

    def __init__(self, num_nodes, num_rels, n_train, n_valid, n_test, seed=0, zipf=0.8):
        rs = np.random.RandomState(seed)
        p = (np.arange(num_nodes) + 1.0) ** (-zipf)
        p /= p.sum()

        def draw(n):
            s = rs.choice(num_nodes, size=n, p=p)
            o = rs.choice(num_nodes, size=n, p=p)
            r = rs.randint(0, num_rels, size=n)
            return np.stack([s, r, o], axis=1).astype(np.int64)

        self.num_nodes, self.num_rels = num_nodes, num_rels
        self.train, self.valid, self.test = draw(n_train), draw(n_valid), draw(n_test)


_REGISTRY = {}


def register_dataset(name, data):
    _REGISTRY[name] = data


def load_data(name):
    return _REGISTRY[name]


def install():
    dgl = types.ModuleType('dgl')
    dgl.DGLGraph = graphs.SimpleGraph
    nn_mod = types.ModuleType('dgl.nn')
    pt = types.ModuleType('dgl.nn.pytorch')
    pt.RelGraphConv = RelGraphConv
    contrib = types.ModuleType('dgl.contrib')
    data = types.ModuleType('dgl.contrib.data')
    data.load_data = load_data
    dgl.nn, nn_mod.pytorch, dgl.contrib, contrib.data = nn_mod, pt, contrib, data
    sys.modules.update({'dgl': dgl, 'dgl.nn': nn_mod, 'dgl.nn.pytorch': pt,
                        'dgl.contrib': contrib, 'dgl.contrib.data': data})
    return dgl
