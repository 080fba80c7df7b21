"""Knowledge-graph datasets.

``load_data(name)`` returns an object with ``num_nodes, num_rels, train, valid, test`` ((n,3) int64
arrays of (subject, relation, object)) as ``dgl.contrib.data.load_data`` did for the reference
(kgvae/link_predict.py:105-110).  Files are read from ``$GCNVAE_DATA`` or ``~/.dgl/<name>/`` in the
DGL-0.4 layout the reference's own code assumes (kgvae/utils.py:249-256): ``entities.dict`` /
``relations.dict`` with ``id<TAB>name`` lines and ``train.txt`` / ``valid.txt`` / ``test.txt`` with
``subject<TAB>relation<TAB>object`` names.  No dataset ships with this repo and there is no network
here, so ``name`` may also be ``synthetic:<entities>:<relations>:<train>[:<valid>:<test>[:<seed>]]``
(Zipf(0.8) endpoints -- the hub skew of FB15k-237 -- and uniform relations).
"""
import os

import numpy as np

FB15K237 = dict(num_nodes=14541, num_rels=237, n_train=272115, n_valid=17535, n_test=20466)
WN18RR = dict(num_nodes=40943, num_rels=11, n_train=86835, n_valid=3034, n_test=3134)


class KGDataset:
    def __init__(self, name, num_nodes, num_rels, train, valid, test):
        self.name, self.num_nodes, self.num_rels = name, int(num_nodes), int(num_rels)
        self.train, self.valid, self.test = train, valid, test


def synthetic_kg(num_nodes, num_rels, n_train, n_valid=0, n_test=0, seed=0, zipf=0.8, name='synthetic'):
    rs = np.random.RandomState(seed)
    p = (np.arange(num_nodes) + 1.0) ** (-zipf)
    p /= p.sum()

    def draw(n):
        s = rs.choice(num_nodes, size=n, p=p)
        o = rs.choice(num_nodes, size=n, p=p)
        r = rs.randint(0, num_rels, size=n)
        return np.stack([s, r, o], axis=1).astype(np.int64)

    return KGDataset(name, num_nodes, num_rels, draw(n_train), draw(n_valid), draw(n_test))


def _read_dict(path):
    out = {}
    with open(path) as f:
        for line in f:
            line = line.rstrip('\n')
            if line:
                idx, name = line.split('\t')
                out[name] = int(idx)
    return out


def _read_triplets(path, ent, rel):
    rows = []
    with open(path) as f:
        for line in f:
            parts = line.rstrip('\n').split('\t')
            if len(parts) == 3:
                rows.append((ent[parts[0]], rel[parts[1]], ent[parts[2]]))
    return np.asarray(rows, dtype=np.int64).reshape(-1, 3)


def load_data(name):
    if name.startswith('synthetic:'):
        f = [int(x) for x in name.split(':')[1:]]
        f += [0] * (6 - len(f))
        return synthetic_kg(f[0], f[1], f[2], f[3], f[4], seed=f[5], name=name)
    if name in ('FB15k-237-synthetic', 'WN18RR-synthetic'):
        cfg = FB15K237 if name.startswith('FB') else WN18RR
        return synthetic_kg(cfg['num_nodes'], cfg['num_rels'], cfg['n_train'], cfg['n_valid'], cfg['n_test'], name=name)
    root = os.environ.get('GCNVAE_DATA', os.path.join(os.path.expanduser('~'), '.dgl'))
    d = os.path.join(root, name)
    if not os.path.isdir(d):
        raise FileNotFoundError(f'dataset directory {d} not found (expected entities.dict, relations.dict, '
                                f'train/valid/test.txt); use "{name}-synthetic" or "synthetic:..." without files')
    ent = _read_dict(os.path.join(d, 'entities.dict'))
    rel = _read_dict(os.path.join(d, 'relations.dict'))
    return KGDataset(name, len(ent), len(rel), *(_read_triplets(os.path.join(d, s + '.txt'), ent, rel)
                                                 for s in ('train', 'valid', 'test')))
