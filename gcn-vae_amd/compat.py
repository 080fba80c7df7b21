"""Drop-in aliases so that code written against the reference's imports finds the gfx950 path.

    python -m gcn_vae_amd.compat /path/to/kgvae/link_predict.py -d FB15k-237 --gpu 0 ...

``install()`` registers, in ``sys.modules``:
    dgl.DGLGraph                 -> gcn_vae_amd.graph.KGraph
    dgl.nn.pytorch.RelGraphConv  -> gcn_vae_amd.layers.RelGraphConv        (K1/K2 kernels)
    dgl.contrib.data.load_data   -> gcn_vae_amd.data.load_data
    model                        -> KGVAE, RGCN, BaseRGCN, EmbeddingLayer, DistLayer   (kgvae/model.py)
    flow_network                 -> MADE, PermuteLayer, MaskedLinear                   (kgvae/flow_network.py)
    utils                        -> sampling + ranking + probability helpers           (kgvae/utils.py)
so ``from dgl.nn.pytorch import RelGraphConv``, ``from model import KGVAE`` and ``import utils`` resolve
to this package.  A script driven this way keeps its own Python (its ``LinkPredict`` scorer then runs on
torch's device ops); the fused decoder/optimiser path is ``gcn_vae_amd.train``.
"""
import runpy
import sys
import types


def install():
    from . import data, encoders, flows, graph, layers, prob, ranking, sampling
    dgl = types.ModuleType('dgl')
    dgl.DGLGraph = graph.KGraph
    nn_mod, pt = types.ModuleType('dgl.nn'), types.ModuleType('dgl.nn.pytorch')
    pt.RelGraphConv = layers.RelGraphConv
    contrib, cdata = types.ModuleType('dgl.contrib'), types.ModuleType('dgl.contrib.data')
    cdata.load_data = data.load_data
    dgl.nn, nn_mod.pytorch, dgl.contrib, contrib.data = nn_mod, pt, contrib, cdata
    model = types.ModuleType('model')
    for name in ('KGVAE', 'RGCN', 'BaseRGCN', 'EmbeddingLayer', 'DistLayer'):
        setattr(model, name, getattr(encoders, name))
    flow_network = types.ModuleType('flow_network')
    for name in ('MADE', 'PermuteLayer', 'MaskedLinear'):
        setattr(flow_network, name, getattr(flows, name))
    utils = types.ModuleType('utils')
    for mod in (sampling, ranking, prob):
        for name in dir(mod):
            if not name.startswith('_') and callable(getattr(mod, name)):
                setattr(utils, name, getattr(mod, name))
    sys.modules.update({'dgl': dgl, 'dgl.nn': nn_mod, 'dgl.nn.pytorch': pt, 'dgl.contrib': contrib,
                        'dgl.contrib.data': cdata, 'model': model, 'flow_network': flow_network, 'utils': utils})
    return dgl


if __name__ == '__main__':
    if len(sys.argv) < 2:
        sys.exit('usage: python -m gcn_vae_amd.compat <script.py> [script args...]')
    install()
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name='__main__')
