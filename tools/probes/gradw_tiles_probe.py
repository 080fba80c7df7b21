#!/usr/bin/env python3
"""The MADE weight-gradient product at WN18RR size on operands that are NOT cache-resident (four operand sets of 164 MB used in turn:
the MALL holds 256 MB): [row][k] operands against 64-deep K tiles, by number of K slices.  us per product (kernel + split sums)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import ops
from tools.microbench import timeit

m = int(sys.argv[1]) if len(sys.argv) > 1 else 40943
d, S, L, SETS = 200, 5, 5, 4
dev = torch.device('cuda:0')
bf = dict(dtype=torch.bfloat16, device=dev)
np8, T = (m + 7) // 8 * 8, (m + 63) // 64
k8, k64 = S * np8, S * T * 64
plain = [torch.randn(2 * d, k8, device=dev).to(torch.bfloat16) for _ in range(SETS)]
tiles = [torch.randn(S * T, L * d, 64, device=dev).to(torch.bfloat16) for _ in range(SETS)]      # production form: a column range of a wider tile
gw, gb = torch.zeros(d, d, device=dev), torch.zeros(d, device=dev)
for split in (96, 128, 192, 256):
    def run_plain():
        for t in plain:
            ops.gemm_bf16_gradw(t[d:], t[:d], d, d, k8, gw, a_rowsum=gb, split_k=split)
    def run_tiles():
        for t in tiles:
            ops.gemm_bf16_gradw_tiles(t[:, d:2 * d], L * d * 64, t[:, :d], L * d * 64, d, d, k64, gw, a_rowsum=gb, split_k=split)
    print(f'm={m} {d}x{d} over {k64}, {split:3d} slices: [row][k] {timeit(run_plain, iters=10) / SETS:6.1f} us, 64-deep tiles {timeit(run_tiles, iters=10) / SETS:6.1f} us', flush=True)
