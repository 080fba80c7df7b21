"""Time the device batch samplers at FB15k-237 size (272 115 triplets, 14 541 entities, sample 30 000 edges) against the host
pipeline (gcn_vae_amd.sampling, the reference-exact numpy statement).  python tools/sampler_bench.py [--host]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_vae_amd import sampling                      # noqa: E402
from gcn_vae_amd.data import FB15K237, synthetic_kg   # noqa: E402
from gcn_vae_amd.device_sampling import DeviceSampler  # noqa: E402


def main():
    data = synthetic_kg(FB15K237['num_nodes'], FB15K237['num_rels'], FB15K237['n_train'], seed=0)
    k = 30000
    for mode in ('uniform', 'neighbor'):
        sm = DeviceSampler(data.train, data.num_nodes, data.num_rels, 'cuda', seed=1, sampler=mode)
        sm.sample(k, 0.5, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            sm.sample(k, 0.5, 10)
        torch.cuda.synchronize()
        print(f'device {mode:9s}: {(time.perf_counter() - t0) / 5 * 1e3:8.2f} ms / batch', flush=True)
    if '--host' in sys.argv:
        adj, deg = sampling.get_adj_and_degrees(data.num_nodes, data.train)
        for mode in ('uniform', 'neighbor'):
            np.random.seed(0)
            t0 = time.perf_counter()
            sampling.generate_sampled_graph_and_labels(data.train, k, 0.5, data.num_rels, adj, deg, 10, mode)
            print(f'host   {mode:9s}: {(time.perf_counter() - t0) * 1e3:8.2f} ms / batch', flush=True)


if __name__ == '__main__':
    main()
