#!/usr/bin/env python3
"""Time gv_made_chain at FB15k-237 size (n = 14541, d = 200, 4 hidden layers): the forward chain and the backward-x chain."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from gcn_vae_amd import ops

dev = torch.device('cuda:0')
n = int(os.environ.get('N', 14541))
d = 200
bf = dict(dtype=torch.bfloat16, device=dev)
widths = [d, 200, 200, 200, 200, 400]
L = len(widths) - 1
ws = [torch.randn(widths[i + 1], widths[i], device=dev) * 0.1 for i in range(L)]
bs = [torch.randn(widths[i + 1], device=dev) for i in range(L)]
pk = [ops.made_pack_weight(w) for w in ws]
x = torch.randn(n, d, device=dev).to(torch.bfloat16)
npad = (n + 7) // 8 * 8
ob = [torch.empty(n, widths[i + 1], **bf) for i in range(L - 1)]
ot = [torch.empty(widths[i + 1], npad, **bf) for i in range(L - 1)]
net = torch.empty(n, 400, device=dev)
g = torch.randn(n, 400, device=dev).to(torch.bfloat16)
gb = [torch.empty(n, widths[i], **bf) for i in range(1, L)]
gt = [torch.empty(widths[i], npad, **bf) for i in range(1, L)]
gold = torch.zeros(n, d, device=dev)


def fwd():
    ops.made_chain(x, n, [dict(w_packed=pk[l][0], n=widths[l + 1], k=widths[l], bias=bs[l], relu=True, out_bf16=ob[l],
                               out_bf16_t=ot[l]) for l in range(L - 1)] +
                   [dict(w_packed=pk[L - 1][0], n=400, k=200, bias=bs[L - 1], out_f32=net)])


def bwd():
    ops.made_chain(g, n, [dict(w_packed=pk[l][1], n=widths[l], k=widths[l + 1], mask=ob[l - 1], out_bf16=gb[l - 1],
                               out_bf16_t=gt[l - 1]) for l in reversed(range(1, L))] +
                   [dict(w_packed=pk[0][1], n=d, k=200, out_f32=gold, accumulate=True)])


for name, fn in (('forward chain', fwd), ('backward chain', bwd)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for _ in range(20):
                fn()
    gr.replay()
    torch.cuda.synchronize()
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f'{name:20s} {e0.elapsed_time(e1) * 1000 / 20:8.2f} us')
