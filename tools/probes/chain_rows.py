"""Does the bf16 MADE node pay for the last, partly filled round of 64-row workgroups?  Forward + backward of one MADE(200, 200, 3)
under bf16 products at row counts around WN18RR's 40 943 (= 639.7 tiles = 2.5 rounds of 256 workgroups): us per call and per 1 000 rows.
    python tools/probes/chain_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from microbench import timeit  # noqa: E402

from gcn_vae_amd import ops  # noqa: E402
from gcn_vae_amd.flows import MADE  # noqa: E402

torch.manual_seed(0)
m = MADE(200, 200, 3).cuda()
for rows in (16384, 24576, 32768, 36864, 40943, 40960, 45056, 49152, 65536):
    z = torch.randn(rows, 200, device='cuda', requires_grad=True)

    def step():
        with ops.gemm_precision('bf16'):
            x, ld = m(z)
            (x.sum() + ld.sum()).backward()
        z.grad = None
        for p in m.parameters():
            p.grad = None
    os.environ['MB_GRAPH'] = '0'
    t = timeit(step, iters=10, warm=3)
    print(f'rows {rows:6d} = {rows / 64 / 256:5.2f} rounds of 256 x 64-row workgroups: {t:8.1f} us per forward + backward, {t / rows * 1e3:6.2f} us per 1000 rows',
          flush=True)
