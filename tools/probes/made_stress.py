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



def run(multi, graph, n, d, steps, precision, zs):
    made.MADE_ROW_BLOCKS = made.MADE_F32_ROW_BLOCKS = 2 if multi else 1
    made.MADE_F32_ROW_BLOCKS_MIN_TILES = made.MADE_ROW_BLOCKS_MIN_TILES = 1
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


def check(n=40943, steps=6, precision='bf16', d=200, verbose=False):
    """Raises AssertionError when any output or gradient of any step differs between the multi-stream and the plain form."""
    keep = (made.MADE_ROW_BLOCKS, made.MADE_F32_ROW_BLOCKS, made.MADE_F32_ROW_BLOCKS_MIN_TILES, made.MADE_ROW_BLOCKS_MIN_TILES,
            ops.BWD_SIDE, made.MADE_PREPARE, made.GRADW_SPLIT_MAX_SIDE)
    made.GRADW_SPLIT_MAX_SIDE = made.GRADW_SPLIT_MAX      # the same K slices on the side stream as on the main one: same sums, same bits
    torch.manual_seed(0)
    zs = [torch.randn(n, d, device='cuda') for _ in range(steps)]
    try:
        for graph in (False, True):
            ref, got = run(False, graph, n, d, steps, precision, zs), run(True, graph, n, d, steps, precision, zs)
            bad = [(i, j) for i, (a, b) in enumerate(zip(ref, got)) for j, (u, v) in enumerate(zip(a, b)) if not torch.equal(u, v)]
            if verbose:
                print(f'n={n} {precision} {"captured step" if graph else "eager steps"}: {steps} steps x {len(ref[0])} tensors, '
                      f'mismatches: {bad[:8] if bad else "none"}', flush=True)
            assert not bad, (graph, bad[:8])
            assert all(float(t.abs().max()) > 0 for t in ref[-1][3:])          # the gradients are there
    finally:
        (made.MADE_ROW_BLOCKS, made.MADE_F32_ROW_BLOCKS, made.MADE_F32_ROW_BLOCKS_MIN_TILES, made.MADE_ROW_BLOCKS_MIN_TILES,
         ops.BWD_SIDE, made.MADE_PREPARE, made.GRADW_SPLIT_MAX_SIDE) = keep


if __name__ == '__main__':
    check(int(sys.argv[1]) if len(sys.argv) > 1 else 40943, int(sys.argv[2]) if len(sys.argv) > 2 else 6,
          sys.argv[3] if len(sys.argv) > 3 else 'bf16', verbose=True)       # precision 'f32': the fp32 node (a launch per product)
    print('ok')
