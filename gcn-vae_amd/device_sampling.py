"""Device-side batch preparation for the mini-batch regime.

The reference prepares every training batch on the host with numpy and python loops
(kgvae/utils.py:79-171: uniform edge sample -> relabel -> negative sampling -> graph split -> reverse
edges -> sort by (dst, src, rel) -> 1/in-degree norm); restated faithfully in ``sampling.py`` that costs
25-30 ms per step here against ~2 ms of GPU compute.  ``DeviceSampler`` does the same pipeline with device
tensor ops (sort / unique / bincount) and hands the model a graph handle whose edges already live on the GPU.

It draws from torch's device generator, NOT from numpy's global stream, so batches differ from the
reference's for a given seed: this is the throughput mode.  ``sampling.generate_sampled_graph_and_labels``
stays the reference-exact path (golden-vector tested).  Only the uniform edge sampler is offered (the
neighbourhood sampler is inherently sequential).
"""
from dataclasses import dataclass

import torch

from .graph import KGraph


@dataclass
class DeviceBatch:
    g: KGraph                  # sub-graph handle (edges on the device, (dst, src, rel)-sorted)
    node_id: torch.Tensor      # (N, 1) int64  global ids of the relabelled nodes
    edge_type: torch.Tensor    # (E,)  int64   relation id (reverse edges: + num_rels)
    edge_norm: torch.Tensor    # (E, 1) fp32   1 / in-degree of the edge's destination
    samples: torch.Tensor      # (T, 3) int64  positives followed by negatives, relabelled ids
    labels: torch.Tensor       # (T,)  fp32


class DeviceSampler:
    def __init__(self, triplets, num_nodes, num_rels, device, seed=None):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('DeviceSampler prepares batches on a ROCm device; use gcn_vae_amd.sampling on the host')
        self.triplets = torch.as_tensor(triplets, dtype=torch.int64).to(self.device)
        self.num_nodes, self.num_rels = int(num_nodes), int(num_rels)
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(int(seed))

    def _sorted_graph(self, n, src, rel, dst):
        """build_graph_from_triplets: add reverse edges, order by (dst, src, rel), 1/in-degree norm."""
        s = torch.cat([src, dst])
        d = torch.cat([dst, src])
        r = torch.cat([rel, rel + self.num_rels])
        key = (d * n + s) * (2 * self.num_rels) + r            # < n^2 * 2R: fits int64 for any realistic n
        order = torch.argsort(key)
        s, d, r = s[order], d[order], r[order]
        # in-degrees from the sorted destination column (torch.bincount would synchronise to size its output)
        bounds = torch.searchsorted(d, torch.arange(n + 1, device=d.device, dtype=d.dtype))
        deg = (bounds[1:] - bounds[:-1]).to(torch.float32)
        norm = torch.where(deg > 0, 1.0 / deg.clamp(min=1.0), torch.zeros_like(deg))
        g = KGraph.from_device_edges(n, s, d, dst_sorted=True)
        return g, r, norm[d].view(-1, 1)

    def sample(self, sample_size, split_size=0.5, negative_rate=10):
        """generate_sampled_graph_and_labels(..., sampler='uniform') on the device."""
        dev, gen = self.device, self.gen
        n_trip = self.triplets.shape[0]
        pick = torch.randperm(n_trip, device=dev, generator=gen)[:sample_size]
        sub = self.triplets[pick]
        src, rel, dst = sub[:, 0], sub[:, 1], sub[:, 2]
        uniq, inv = torch.unique(torch.cat([src, dst]), return_inverse=True)     # sorted ids, like np.unique
        src, dst = inv[:sample_size], inv[sample_size:]
        n = int(uniq.numel())                                                     # one host sync per batch
        pos = torch.stack([src, rel, dst], dim=1)
        # negative_sampling: corrupt subject or object (coin > 0.5 -> subject) with a uniform entity
        total = sample_size * negative_rate
        neg = pos.repeat(negative_rate, 1)
        values = torch.randint(0, n, (total,), device=dev, generator=gen)
        hit_subject = torch.rand(total, device=dev, generator=gen) > 0.5
        neg[:, 0] = torch.where(hit_subject, values, neg[:, 0])
        neg[:, 2] = torch.where(hit_subject, neg[:, 2], values)
        samples = torch.cat([pos, neg])
        labels = torch.zeros(sample_size * (negative_rate + 1), dtype=torch.float32, device=dev)
        labels[:sample_size] = 1
        # graph split: a random part of the sampled edges forms the message-passing graph
        keep = torch.randperm(sample_size, device=dev, generator=gen)[:int(sample_size * split_size)]
        g, etype, enorm = self._sorted_graph(n, src[keep], rel[keep], dst[keep])
        return DeviceBatch(g, uniq.view(-1, 1), etype, enorm, samples, labels)

    def full_graph(self, triplets=None):
        """build_test_graph on the device: all triplets, identity node ids."""
        t = self.triplets if triplets is None else torch.as_tensor(triplets, dtype=torch.int64).to(self.device)
        g, etype, enorm = self._sorted_graph(self.num_nodes, t[:, 0], t[:, 1], t[:, 2])
        node_id = torch.arange(self.num_nodes, dtype=torch.int64, device=self.device).view(-1, 1)
        return g, node_id, etype, enorm
