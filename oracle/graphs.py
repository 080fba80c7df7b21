"""Oracle restatement of the host-side graph pipeline  --  TEST INFRASTRUCTURE (pinned by golden vectors).

Follows /root/reference/kgvae/utils.py:
  get_adj_and_degrees               :20-30
  sample_edge_neighborhood          :33-76
  sample_edge_uniform               :79-82
  generate_sampled_graph_and_labels :85-124
  comp_deg_norm                     :127-132
  build_graph_from_triplets         :135-150   (reverse edges, sort by (dst, src, rel))
  build_test_graph                  :153-155
  negative_sampling                 :158-171
and /root/reference/kgvae/link_predict.py:95-100 (node_norm_to_edge_norm).

numpy's global RNG is consumed in exactly the reference's order so that a fixed
``np.random.seed`` reproduces the reference's batches.  ``SimpleGraph`` is the minimal
stand-in for the DGL-0.4 ``DGLGraph`` surface the reference touches.
"""
import numpy as np
import torch


class _EdgeBatch:
    def __init__(self, g):
        self.src = {k: v[g._src] for k, v in g.ndata.items()}
        self.dst = {k: v[g._dst] for k, v in g.ndata.items()}
        self.data = g.edata


class SimpleGraph:
    """add_nodes / add_edges / in_degrees / number_of_nodes / number_of_edges / local_var /
    ndata / edata / apply_edges / __len__  (kgvae/utils.py:127-150, kgvae/link_predict.py:95-100, :216)."""

    def __init__(self):
        self._n = 0
        self._src = torch.zeros(0, dtype=torch.int64)
        self._dst = torch.zeros(0, dtype=torch.int64)
        self.ndata = {}
        self.edata = {}

    def add_nodes(self, n):
        self._n += int(n)

    def add_edges(self, src, dst):
        self._src = torch.cat([self._src, torch.as_tensor(np.asarray(src), dtype=torch.int64)])
        self._dst = torch.cat([self._dst, torch.as_tensor(np.asarray(dst), dtype=torch.int64)])

    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self._src.shape[0])

    def __len__(self):
        return self._n

    def edges(self):
        return self._src, self._dst

    def in_degrees(self, nodes=None):
        deg = torch.bincount(self._dst, minlength=self._n)
        return deg if nodes is None else deg[torch.as_tensor(list(nodes), dtype=torch.int64)]

    def local_var(self):
        g = SimpleGraph()
        g._n, g._src, g._dst = self._n, self._src, self._dst
        g.ndata, g.edata = dict(self.ndata), dict(self.edata)
        return g

    def apply_edges(self, fn):
        self.edata.update(fn(_EdgeBatch(self)))


def get_adj_and_degrees(num_nodes, triplets):
    adj = [[] for _ in range(num_nodes)]
    for i, (s, _, o) in enumerate(triplets):
        adj[s].append([i, o])
        adj[o].append([i, s])
    degrees = np.array([len(a) for a in adj])
    return [np.array(a) for a in adj], degrees


def sample_edge_uniform(adj_list, degrees, n_triplets, sample_size):
    return np.random.choice(np.arange(n_triplets), sample_size, replace=False)


def sample_edge_neighborhood(adj_list, degrees, n_triplets, sample_size):
    edges = np.zeros((sample_size), dtype=np.int32)
    remaining = np.array([d for d in degrees])
    picked = np.array([False for _ in range(n_triplets)])
    seen = np.array([False for _ in degrees])
    for i in range(sample_size):
        weights = remaining * seen
        if np.sum(weights) == 0:
            weights = np.ones_like(weights)
            weights[np.where(remaining == 0)] = 0
        prob = weights / np.sum(weights)
        v = np.random.choice(np.arange(degrees.shape[0]), p=prob)
        nbrs = adj_list[v]
        seen[v] = True
        pick = nbrs[np.random.choice(np.arange(nbrs.shape[0]))]
        while picked[pick[0]]:
            pick = nbrs[np.random.choice(np.arange(nbrs.shape[0]))]
        edges[i] = pick[0]
        picked[pick[0]] = True
        remaining[v] -= 1
        remaining[pick[1]] -= 1
        seen[pick[1]] = True
    return edges


def negative_sampling(pos_samples, num_entity, negative_rate):
    n = len(pos_samples)
    total = n * negative_rate
    neg = np.tile(pos_samples, (negative_rate, 1))
    labels = np.zeros(n * (negative_rate + 1), dtype=np.float32)
    labels[:n] = 1
    values = np.random.randint(num_entity, size=total)
    coin = np.random.uniform(size=total)
    corrupt_subj = coin > 0.5
    corrupt_obj = coin <= 0.5
    neg[corrupt_subj, 0] = values[corrupt_subj]
    neg[corrupt_obj, 2] = values[corrupt_obj]
    return np.concatenate((pos_samples, neg)), labels


def comp_deg_norm(g):
    in_deg = g.in_degrees(range(g.number_of_nodes())).float().numpy()
    with np.errstate(divide='ignore'):
        norm = 1.0 / in_deg
    norm[np.isinf(norm)] = 0
    return norm


def build_graph_from_triplets(num_nodes, num_rels, triplets, graph_cls=SimpleGraph):
    g = graph_cls()
    g.add_nodes(num_nodes)
    src, rel, dst = triplets
    src, dst = np.concatenate((src, dst)), np.concatenate((dst, src))
    rel = np.concatenate((rel, rel + num_rels))
    order = sorted(zip(dst, src, rel))
    dst, src, rel = np.array(order).transpose()
    g.add_edges(src, dst)
    return g, rel, comp_deg_norm(g)


def build_test_graph(num_nodes, num_rels, edges, graph_cls=SimpleGraph):
    src, rel, dst = np.array(edges).transpose()
    return build_graph_from_triplets(num_nodes, num_rels, (src, rel, dst), graph_cls)


def generate_sampled_graph_and_labels(triplets, sample_size, split_size, num_rels, adj_list, degrees,
                                      negative_rate, sampler="uniform", graph_cls=SimpleGraph):
    if sampler == "uniform":
        picked = sample_edge_uniform(adj_list, degrees, len(triplets), sample_size)
    elif sampler == "neighbor":
        picked = sample_edge_neighborhood(adj_list, degrees, len(triplets), sample_size)
    else:
        raise ValueError("Sampler type must be either 'uniform' or 'neighbor'.")
    sub = triplets[picked]
    src, rel, dst = np.array(sub).transpose()
    uniq_v, inverse = np.unique((src, dst), return_inverse=True)
    src, dst = np.reshape(inverse, (2, -1))
    relabeled = np.stack((src, rel, dst)).transpose()
    samples, labels = negative_sampling(relabeled, len(uniq_v), negative_rate)
    n_graph = int(sample_size * split_size)
    keep = np.random.choice(np.arange(sample_size), size=n_graph, replace=False)
    g, rel, norm = build_graph_from_triplets(len(uniq_v), num_rels, (src[keep], rel[keep], dst[keep]),
                                             graph_cls)
    return g, uniq_v, rel, norm, samples, labels


def node_norm_to_edge_norm(g, node_norm):
    """link_predict.py:95-100 -- edge norm = node norm of the edge's destination, shape (E, 1)."""
    g = g.local_var()
    g.ndata['norm'] = node_norm
    g.apply_edges(lambda edges: {'norm': edges.dst['norm']})
    return g.edata['norm']
