#!/usr/bin/env python3
"""Self-consistency of the multi-stream MADE paths at WN18RR size: the bf16 node with two row blocks, side-stream weight gradients
and prepared parameters against the same node with all of that switched off -- outputs and every gradient bit for bit, over
several eager steps and over replays of a captured step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import made, ops
from gcn_vae_amd.flows import MADE
from gcn_vae_amd.optim import FlatAdam

n, d, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40943, 200, int(sys.argv[2]) if len(sys.argv) > 2 else 6
precision = sys.argv[3] if len(sys.argv) > 3 else 'bf16'          # 'f32': the fp32 node (a launch per product)
torch.manual_seed(0)
zs = [torch.randn(n, d, device='cuda') for _ in range(steps)]


def run(multi, graph):
    made.MADE_ROW_BLOCKS = made.MADE_F32_ROW_BLOCKS = 2 if multi else 1
    made.MADE_F32_ROW_BLOCKS_MIN_TILES = 1
    ops.BWD_SIDE = multi
    made.MADE_PREPARE = multi
    torch.manual_seed(1)
    ms = [MADE(d, d, 3).cuda() for _ in range(2)]                      # two flows in a row: the second one's backward runs beside the first one's products
    opt = FlatAdam([p for m in ms for p in m.parameters()], lr=1e-3, max_grad_norm=1.0)
    out = []
    zin = torch.empty(n, d, device='cuda')

    def step():
        opt.zero_grad()
        with ops.gemm_precision(precision):
            ops.made_prepare([m.call_arguments() for m in ms])
            x = zin.clone().requires_grad_(True)
            y = x
            ld = 0
            for m in ms:
                y, l = m(y)
                ld = ld + l
            ops.made_prepare_finish()
            loss = (y * y).mean() + ld.mean()
            loss.backward()
        return loss, y, x.grad
    if graph:
        zin.copy_(zs[0])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            res = step()
    for i in range(steps):
        zin.copy_(zs[i])
        if graph:
            g.replay()
        else:
            res = step()
        torch.cuda.synchronize()
        out.append([t.detach().clone() for t in res] + [p.grad.detach().clone() for m in ms for p in m.parameters()])
    opt.close()
    return out


for graph in (False, True):
    ref, got = run(False, graph), run(True, graph)
    bad = [(i, j) for i, (a, b) in enumerate(zip(ref, got)) for j, (u, v) in enumerate(zip(a, b)) if not torch.equal(u, v)]
    print(f'n={n} {precision} {"captured step" if graph else "eager steps"}: {steps} steps x {len(ref[0])} tensors, mismatches: {bad[:8] if bad else "none"}', flush=True)
    assert not bad
print('ok')
