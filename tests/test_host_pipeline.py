"""Host logic of the product (no GPU): graph construction and sampling reproduce the reference's
numpy RNG stream (golden vectors), the graph handle behaves like the DGL surface, the data reader
parses the DGL-0.4 on-disk layout."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden


def test_sampling_matches_reference_vectors():
    from gcn_vae_amd import sampling
    g = load_golden('pipeline.npz')
    train = g['train'].numpy()
    adj, deg = sampling.get_adj_and_degrees(300, train)
    assert np.array_equal(deg, g['degrees'].numpy())
    assert np.array_equal(np.concatenate([a.reshape(-1, 2) for a in adj if a.size]), g['adj_flat'].numpy())
    np.random.seed(0)
    ns, nl = sampling.negative_sampling(g['neg_pos'].numpy().copy(), 300, 3)
    assert np.array_equal(ns, g['neg_samples'].numpy()) and np.array_equal(nl, g['neg_labels'].numpy())
    np.random.seed(1)
    assert np.array_equal(sampling.sample_edge_uniform(adj, deg, len(train), 100), g['uniform_edges'].numpy())
    np.random.seed(2)
    assert np.array_equal(sampling.sample_edge_neighborhood(adj, deg, len(train), 60), g['neighbor_edges'].numpy())
    for tag, sampler, seed in (('u', 'uniform', 3), ('n', 'neighbor', 4)):
        np.random.seed(seed)
        gr, uniq_v, rel, norm, samples, labels = sampling.generate_sampled_graph_and_labels(
            train, 200, 0.5, 12, adj, deg, 4, sampler)
        src, dst = gr.edges()
        assert torch.equal(src, g[f'{tag}_src']) and torch.equal(dst, g[f'{tag}_dst'])
        assert np.array_equal(uniq_v, g[f'{tag}_uniq_v'].numpy()) and np.array_equal(rel, g[f'{tag}_rel'].numpy())
        assert np.array_equal(norm, g[f'{tag}_norm'].numpy())
        assert np.array_equal(samples, g[f'{tag}_samples'].numpy()) and np.array_equal(labels, g[f'{tag}_labels'].numpy())
        en = sampling.node_norm_to_edge_norm(gr, torch.from_numpy(norm).view(-1, 1))
        assert torch.equal(en, g[f'{tag}_edge_norm'])
    tg, trel, tnorm = sampling.build_test_graph(300, 12, g['valid'].numpy())
    ts, td = tg.edges()
    assert torch.equal(ts, g['test_src']) and torch.equal(td, g['test_dst'])
    assert np.array_equal(trel, g['test_rel'].numpy()) and np.array_equal(tnorm, g['test_norm'].numpy())
    with pytest.raises(ValueError, match="'uniform' or 'neighbor'"):
        sampling.generate_sampled_graph_and_labels(train, 10, 0.5, 12, adj, deg, 1, 'nope')


def test_graph_handle_surface():
    from gcn_vae_amd.graph import KGraph
    g = KGraph()
    g.add_nodes(5)
    g.add_edges([0, 1, 1, 4], [1, 2, 2, 0])
    assert len(g) == 5 and g.number_of_nodes() == 5 and g.number_of_edges() == 4
    assert g.in_degrees(range(5)).tolist() == [1, 1, 2, 0, 0]
    loc = g.local_var()
    loc.ndata['norm'] = torch.arange(5.0).view(-1, 1)
    loc.apply_edges(lambda edges: {'norm': edges.dst['norm']})
    assert loc.edata['norm'].view(-1).tolist() == [1.0, 2.0, 2.0, 0.0]
    assert 'norm' not in g.ndata and 'norm' not in g.edata
    with pytest.raises(ValueError):
        g.add_edges([0], [7])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        g.device_index('cpu')


def test_data_reader_dgl_layout(tmp_path, monkeypatch):
    from gcn_vae_amd import data
    d = tmp_path / 'toy'
    d.mkdir()
    (d / 'entities.dict').write_text('0\t/m/a\n1\t/m/b\n2\t/m/c\n')
    (d / 'relations.dict').write_text('0\tlikes\n1\tknows\n')
    (d / 'train.txt').write_text('/m/a\tlikes\t/m/b\n/m/b\tknows\t/m/c\n')
    (d / 'valid.txt').write_text('/m/c\tlikes\t/m/a\n')
    (d / 'test.txt').write_text('')
    monkeypatch.setenv('GCNVAE_DATA', str(tmp_path))
    ds = data.load_data('toy')
    assert (ds.num_nodes, ds.num_rels) == (3, 2)
    assert ds.train.tolist() == [[0, 0, 1], [1, 1, 2]] and ds.valid.tolist() == [[2, 0, 0]] and ds.test.shape == (0, 3)
    syn = data.load_data('synthetic:50:4:300:20:10:7')
    assert syn.train.shape == (300, 3) and syn.train[:, 1].max() < 4 and syn.train[:, [0, 2]].max() < 50
    with pytest.raises(FileNotFoundError):
        data.load_data('FB15k-237')


def test_modules_construct_and_reject_cpu():
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.graph import KGraph
    from gcn_vae_amd.layers import RelGraphConv
    from gcn_vae_amd.train import LinkPredict, build_parser
    with pytest.raises(ValueError, match='multiplier of num_bases'):
        RelGraphConv(200, 200, 22, 'bdd', 100)        # C3: --n-bases 100 is clamped to 22 relations
    with pytest.raises(ValueError, match="'basis' or 'bdd'"):
        RelGraphConv(8, 8, 4, 'nope', 2)
    net = LinkPredict(KGVAE, 30, 8, 3, num_bases=2, num_hidden_layers=2, k=2, n_flows=1)
    g = KGraph()
    g.add_nodes(30)
    g.add_edges([0, 1], [1, 2])
    with pytest.raises(RuntimeError, match='no CPU'):
        net(g, torch.arange(30).view(-1, 1), torch.tensor([0, 1]), torch.ones(2, 1))
    args = build_parser().parse_args(['-d', 'x'])
    ref_defaults = dict(dropout=0.2, n_hidden=500, gpu=-1, lr=1e-3, n_bases=100, n_layers=2, n_epochs=1e5,
                        eval_batch_size=400, regularization=0.01, kl_param=1e-5, mmd_param=0, mog_k=10, n_flows=0,
                        grad_norm=1.0, graph_batch_size=20000, graph_split_size=0.5, negative_sample=10,
                        evaluate_every=200, edge_sampler='uniform', test_mode=False,
                        model_state_file='model_state.pth', model_class='KGVAE', load=False, generate=False)
    for k, v in ref_defaults.items():
        assert getattr(args, k) == v, k


def test_compat_install_aliases_and_state_dict_manifest():
    """``gcn_vae_amd.compat.install()`` -- the route by which the reference's own ``link_predict.py`` finds this package:
    every module name the reference imports (kgvae/model.py:4-8, kgvae/link_predict.py:22-27, kgvae/utils.py:10) resolves
    to the gfx950 classes, and models built THROUGH the aliases have the reference's state_dict keys and shapes
    (tests/golden/state_dict_manifest.json, captured from the reference)."""
    import importlib
    import json
    import os
    import sys
    from gcn_vae_amd import compat, data, encoders, flows, graph, layers, train
    names = ('dgl', 'dgl.nn', 'dgl.nn.pytorch', 'dgl.contrib', 'dgl.contrib.data', 'model', 'flow_network', 'utils')
    saved = {k: sys.modules.get(k) for k in names}
    try:
        compat.install()
        assert importlib.import_module('dgl').DGLGraph is graph.KGraph
        assert importlib.import_module('dgl.nn.pytorch').RelGraphConv is layers.RelGraphConv
        assert importlib.import_module('dgl.contrib.data').load_data is data.load_data
        model = importlib.import_module('model')
        assert model.KGVAE is encoders.KGVAE and model.RGCN is encoders.RGCN and model.BaseRGCN is encoders.BaseRGCN
        fn = importlib.import_module('flow_network')
        assert fn.MADE is flows.MADE and fn.PermuteLayer is flows.PermuteLayer and fn.MaskedLinear is flows.MaskedLinear
        utils = importlib.import_module('utils')
        for f in ('get_adj_and_degrees', 'generate_sampled_graph_and_labels', 'build_test_graph', 'calc_mrr',
                  'gaussian_parameters', 'sample_gaussian', 'log_normal', 'log_normal_mixture'):
            assert callable(getattr(utils, f)), f
        here = os.path.dirname(os.path.abspath(__file__))
        manifest = json.load(open(os.path.join(here, 'golden', 'state_dict_manifest.json')))
        for n_flows in (0, 3):      # the constructor call of kgvae/link_predict.py:123-135 at the fixtures' C1 size
            net = train.LinkPredict(model.KGVAE, 1000, 16, 20, num_bases=4, num_hidden_layers=2, dropout=0.2, use_cuda=False,
                                    reg_param=0.01, kl_param=1e-5, mmd_param=1.0, k=10, n_flows=n_flows)
            got = {k: list(v.shape) for k, v in net.state_dict().items()}
            assert got == manifest['LinkPredict(KGVAE,n_flows=%d)' % n_flows]
        rg = model.RGCN(50, 8, 8, 6, 2, num_hidden_layers=2, dropout=0.0, use_self_loop=True, use_cuda=False)
        assert {k: list(v.shape) for k, v in rg.state_dict().items()} == manifest['RGCN(num_hidden_layers=2)']
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_neighborhood_sampler_with_injected_draws_has_the_reference_distribution():
    """sampling.sample_edge_neighborhood_draws (the integer-CDF statement the device kernel reproduces) against
    sampling.sample_edge_neighborhood (numpy's stream, pinned to the reference's golden vector above): the same
    distribution over pick sequences on a small graph, and the same invariants."""
    from collections import Counter

    from gcn_vae_amd import sampling
    trip = np.array([[0, 0, 1], [1, 0, 2], [2, 1, 0], [3, 0, 4], [4, 1, 4], [1, 1, 3], [5, 0, 0], [2, 0, 3]])
    n, k, trials = 6, 3, 6000
    adj_list, deg = sampling.get_adj_and_degrees(n, trip)
    csr = sampling.adjacency_csr(n, trip)
    assert np.array_equal(csr[3], deg) and all(np.array_equal(np.stack([csr[1], csr[2]], 1)[csr[0][v]:csr[0][v + 1]], adj_list[v])
                                               for v in range(n) if deg[v])
    np.random.seed(0)
    ref = Counter(tuple(sampling.sample_edge_neighborhood(adj_list, deg, len(trip), k).tolist()) for _ in range(trials))
    rs = np.random.RandomState(1)
    got = Counter()
    for _ in range(trials):
        u = rs.randint(0, 2 ** 32, size=(k, 64), dtype=np.uint64)
        got[tuple(sampling.sample_edge_neighborhood_draws(*csr, len(trip), k, lambda i, a: int(u[i, a])).tolist())] += 1
    for seq in set(ref) | set(got):
        assert len(set(seq)) == k
        p, q = ref[seq] / trials, got[seq] / trials
        assert abs(p - q) < 4.0 * np.sqrt(max(p, q, 1e-3) / trials) + 2e-3, (seq, p, q)
    # exhausting every triplet: all picked once, then -1
    u = rs.randint(0, 2 ** 32, size=(len(trip) + 2, 4200), dtype=np.uint64)
    full = sampling.sample_edge_neighborhood_draws(*csr, len(trip), len(trip) + 2, lambda i, a: int(u[i, a]))
    assert sorted(full[:len(trip)].tolist()) == list(range(len(trip))) and (full[len(trip):] == -1).all()


def test_bench_exits_nonzero_on_a_nonfinite_loss_and_headlines_the_dominant_kernel():
    """bench.py: a result line whose last timed step ended on a NaN / inf loss is not a training measurement -- the program's
    exit status says so; and the line's ``roofline`` names the kernel that is dominant by time (K4 when IAF blocks are on)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.result_exit_code({'loss_is_finite': True}) == 0
    assert bench.result_exit_code({'loss_is_finite': False}) == bench.EXIT_NONFINITE_LOSS != 0
    k1 = {'kernel': 'agg_N_10x20_nb20', 'frac': 0.4}
    detail = {'agg_N_10x20_nb20': {'avg_us': 50.0, 'launches': 2}, 'gradw_10x20_nb20': {'avg_us': 60.0, 'launches': 2}}
    assert bench.dominant_roofline(k1, detail, {}) == (k1, None)
    k4 = {'madechain_fwd': {'avg_us': 80.0, 'launches': 30, 'achieved_TFLOPs': 240.0, 'peak_TFLOPs': 2500.0, 'frac': 0.096,
                            'GFLOP': 19.7, 'operands': 'bf16'},
          'madechain_bwd': {'avg_us': 90.0, 'launches': 30, 'achieved_TFLOPs': 210.0, 'peak_TFLOPs': 2500.0, 'frac': 0.084,
                            'GFLOP': 19.7, 'operands': 'bf16'}}
    roof, roof_k1 = bench.dominant_roofline(k1, detail, k4)
    assert roof['kernel'] == 'madechain_bwd' and roof['bound'] == 'mfma' and roof['unit'] == 'TFLOP/s' and roof_k1 == k1
    assert abs(roof['frac'] - 0.084) < 1e-9 and roof['peak'] == 2500.0


def test_every_environment_knob_is_in_designs_table_and_the_other_way_round():
    """DESIGN.md's knob table against the code: every GV_* environment variable the package, the kernels' host code or bench.py read
    is a row of the table, and every variable the table names is read somewhere."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = (glob.glob(os.path.join(root, 'gcn-vae_amd', '*.py')) + glob.glob(os.path.join(root, 'gcn-vae_amd', 'csrc', '*.hip')) +
             glob.glob(os.path.join(root, 'gcn-vae_amd', 'csrc', '*.h')) + [os.path.join(root, 'bench.py')])
    pat = re.compile(r"""(?:environ(?:\.get|\.setdefault)?[\(\[]\s*['"]|getenv\(")(GV_[A-Z0-9_]+)""")
    in_code = set()
    for f in files:
        in_code |= set(pat.findall(open(f).read()))
    text = open(os.path.join(root, 'DESIGN.md')).read()
    table = text[text.index('## Knobs'):text.index('## Out of scope')]
    in_doc = set(re.findall(r'`(GV_[A-Z0-9_]+)`', table))
    assert in_code, 'no knobs found: the pattern is out of date'
    assert in_code - in_doc == set(), f'read by the code but missing from DESIGN.md: {sorted(in_code - in_doc)}'
    assert in_doc - in_code == set(), f'in DESIGN.md but read nowhere: {sorted(in_doc - in_code)}'
