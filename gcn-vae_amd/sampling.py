"""Host-side graph construction and sampling with the reference's function names
(kgvae/utils.py:20-171; kgvae/link_predict.py:95-100).

numpy's global RNG is consumed in the reference's order (``choice`` for the edge sample,
``randint`` + ``uniform`` for the negatives, ``choice`` for the graph split), so a fixed
``np.random.seed`` reproduces the reference's batches bit for bit (tests/test_host_pipeline.py
checks that against vectors captured from the reference).  Work the reference does in python
loops (adjacency lists, the (dst, src, rel) sort) is vectorised here.
"""
import numpy as np
import torch

from .graph import KGraph


def adjacency_csr(num_nodes, triplets):
    """The incidence lists of get_adj_and_degrees as three flat int32 arrays: vertex v's entries are
    [adj_ptr[v], adj_ptr[v + 1]) of (adj_edge = triplet id, adj_other = other endpoint), in triplet order (subject entry
    before object entry for a self loop); degrees = np.diff(adj_ptr).  This is what the device sampler uploads."""
    triplets = np.asarray(triplets)
    ids = np.arange(len(triplets))
    owner = np.stack([triplets[:, 0], triplets[:, 2]], 1).reshape(-1)
    other = np.stack([triplets[:, 2], triplets[:, 0]], 1).reshape(-1)
    tid = np.repeat(ids, 2)
    order = np.argsort(owner, kind='stable')
    degrees = np.bincount(owner, minlength=num_nodes)
    adj_ptr = np.concatenate([[0], np.cumsum(degrees)])
    return adj_ptr.astype(np.int32), tid[order].astype(np.int32), other[order].astype(np.int32), degrees.astype(np.int32)


def get_adj_and_degrees(num_nodes, triplets):
    """adj_list[v] = array of [triplet id, other endpoint] in triplet order (subject entry before object
    entry for a self loop); degrees[v] = len(adj_list[v])."""
    bounds, tid, other, degrees = adjacency_csr(num_nodes, triplets)
    pairs = np.stack([tid, other], 1).astype(np.int64)
    degrees = degrees.astype(np.int64)
    adj_list = [pairs[bounds[v]:bounds[v + 1]] if degrees[v] else np.array([]) for v in range(num_nodes)]
    return adj_list, degrees


def sample_edge_uniform(adj_list, degrees, n_triplets, sample_size):
    return np.random.choice(np.arange(n_triplets), sample_size, replace=False)


def sample_edge_neighborhood(adj_list, degrees, n_triplets, sample_size):
    """Neighbourhood-expansion sampler; inherently sequential (each draw conditions the next)."""
    edges = np.zeros(sample_size, dtype=np.int32)
    budget = np.array(degrees).copy()
    picked = np.zeros(n_triplets, dtype=bool)
    seen = np.zeros(len(degrees), dtype=bool)
    vertices = np.arange(len(degrees))
    for i in range(sample_size):
        w = budget * seen
        if np.sum(w) == 0:
            w = np.ones_like(w)
            w[np.where(budget == 0)] = 0
        v = np.random.choice(vertices, p=w / np.sum(w))
        nbrs = adj_list[v]
        seen[v] = True
        while True:
            cand = nbrs[np.random.choice(np.arange(nbrs.shape[0]))]
            if not picked[cand[0]]:
                break
        edges[i] = cand[0]
        picked[cand[0]] = True
        budget[v] -= 1
        budget[cand[1]] -= 1
        seen[cand[1]] = True
    return edges


NEIGHBOR_MAX_ATTEMPTS = 4096    # rejection draws per edge before the first unpicked entry of the list is taken


def sample_edge_neighborhood_draws(adj_ptr, adj_edge, adj_other, degrees, n_triplets, sample_size, draw):
    """sample_edge_neighborhood with the random source handed in: ``draw(i, attempt) -> uint32``.

    The same distribution as the reference's sampler (vertex ~ budget * seen, or uniform over the vertices with budget
    left when nothing seen has any; edge uniform over the vertex's unpicked incidence entries by rejection), stated on
    integers so a device kernel can reproduce it bit for bit: attempt 0 selects the vertex at position
    floor(u * W / 2^32) of the integer weight CDF (W = total weight, vertices in id order), attempts >= 1 select the
    incidence entry floor(u * degree / 2^32).  csrc/k_sample.hip k_neighborhood_sample is this loop on the device;
    tests/test_gpu_ops.py holds the two equal draw for draw."""
    edges = np.full(sample_size, -1, dtype=np.int32)
    budget = np.asarray(degrees, dtype=np.int64).copy()
    picked = np.zeros(n_triplets, dtype=bool)
    seen = np.zeros(len(budget), dtype=bool)
    for i in range(sample_size):
        w = budget * seen
        total = int(w.sum())
        if total == 0:
            w = (budget > 0).astype(np.int64)
            total = int(w.sum())
            if total == 0:
                break
        pos = (int(draw(i, 0)) * total) >> 32
        v = int(np.searchsorted(np.cumsum(w), pos, side='right'))
        seen[v] = True
        a0, deg = int(adj_ptr[v]), int(adj_ptr[v + 1] - adj_ptr[v])
        attempt = 1
        while True:
            if attempt > NEIGHBOR_MAX_ATTEMPTS:
                j = int(np.flatnonzero(~picked[adj_edge[a0:a0 + deg]])[0])
            else:
                j = (int(draw(i, attempt)) * deg) >> 32
            attempt += 1
            if not picked[adj_edge[a0 + j]]:
                break
        e, other = int(adj_edge[a0 + j]), int(adj_other[a0 + j])
        edges[i] = e
        picked[e] = True
        budget[v] -= 1
        budget[other] -= 1
        seen[other] = True
    return edges


def negative_sampling(pos_samples, num_entity, negative_rate):
    n = len(pos_samples)
    total = n * negative_rate
    neg = np.tile(pos_samples, (negative_rate, 1))
    labels = np.zeros(n * (negative_rate + 1), dtype=np.float32)
    labels[:n] = 1
    values = np.random.randint(num_entity, size=total)
    coin = np.random.uniform(size=total)
    hit_subject = coin > 0.5
    neg[hit_subject, 0] = values[hit_subject]
    neg[~hit_subject, 2] = values[~hit_subject]
    return np.concatenate((pos_samples, neg)), labels


def comp_deg_norm(g):
    in_deg = g.in_degrees(range(g.number_of_nodes())).float().numpy()
    norm = np.zeros_like(in_deg)
    np.divide(1.0, in_deg, out=norm, where=in_deg > 0)
    return norm


def build_graph_from_triplets(num_nodes, num_rels, triplets):
    """Bidirectional graph (reverse edges carry relation id + num_rels), edges ordered by (dst, src, rel)."""
    src, rel, dst = (np.asarray(a) for a in triplets)
    src, dst = np.concatenate((src, dst)), np.concatenate((dst, src))
    rel = np.concatenate((rel, rel + num_rels))
    order = np.lexsort((rel, src, dst))
    src, dst, rel = src[order].astype(np.int64), dst[order].astype(np.int64), rel[order].astype(np.int64)
    g = KGraph()
    g.add_nodes(num_nodes)
    g.add_edges(src, dst)
    return g, rel, comp_deg_norm(g)


def build_test_graph(num_nodes, num_rels, edges):
    src, rel, dst = np.array(edges).transpose()
    return build_graph_from_triplets(num_nodes, num_rels, (src, rel, dst))


def generate_sampled_graph_and_labels(triplets, sample_size, split_size, num_rels, adj_list, degrees,
                                      negative_rate, sampler="uniform"):
    if sampler == "uniform":
        chosen = sample_edge_uniform(adj_list, degrees, len(triplets), sample_size)
    elif sampler == "neighbor":
        chosen = sample_edge_neighborhood(adj_list, degrees, len(triplets), sample_size)
    else:
        raise ValueError("Sampler type must be either 'uniform' or 'neighbor'.")
    sub = np.asarray(triplets)[chosen]
    src, rel, dst = sub[:, 0], sub[:, 1], sub[:, 2]
    uniq_v, relabel = np.unique((src, dst), return_inverse=True)
    src, dst = np.reshape(relabel, (2, -1))
    samples, labels = negative_sampling(np.stack((src, rel, dst)).transpose(), len(uniq_v), negative_rate)
    keep = np.random.choice(np.arange(sample_size), size=int(sample_size * split_size), replace=False)
    g, rel, norm = build_graph_from_triplets(len(uniq_v), num_rels, (src[keep], rel[keep], dst[keep]))
    return g, uniq_v, rel, norm, samples, labels


def node_norm_to_edge_norm(g, node_norm):
    """(E, 1) edge norm = node norm of each edge's destination (kgvae/link_predict.py:95-100)."""
    _, dst = g.edges()
    return node_norm[dst.to(node_norm.device)]
