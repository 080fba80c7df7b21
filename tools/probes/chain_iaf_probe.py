#!/usr/bin/env python3
"""One forward MADE pass at WN18RR size as a chain with the IAF update fused (ablations through iaf debug bits) against
chain + separate update kernel: us per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import ops, lib
from gcn_vae_amd.lib import ptr
from tools.microbench import timeit

m = int(sys.argv[1]) if len(sys.argv) > 1 else 41143
d, L = 200, 5
dev = torch.device('cuda:0')
widths = [d] * L + [2 * d]
x_old = torch.randn(m, d, device=dev); xb = x_old.to(torch.bfloat16); z = torch.randn(m, d, device=dev)
ws = [torch.randn(widths[i + 1], widths[i], device=dev) / widths[i] ** 0.5 for i in range(L)]
bs = [torch.randn(widths[i + 1], device=dev) * 0.1 for i in range(L)]
cc = torch.ones(d, dtype=torch.int32, device=dev); cc[-1] = 0
mp = (m + 7) // 8 * 8
bf = dict(dtype=torch.bfloat16, device=dev)
acts_b = [torch.empty(m, d, **bf) for _ in range(L - 1)]
acts_t = [torch.empty(d, mp, **bf) for _ in range(L - 1)]
packed = ops.made_pack_weights(ws)
packed_iaf = ops.made_pack_weights(ws, iaf_last=True)
hidden = lambda copies: [dict(w_packed=packed[i][0], n=d, k=d, bias=bs[i], relu=True, **(dict(out_bf16=acts_b[i], out_bf16_t=acts_t[i]) if copies else {}))
                         for i in range(L - 1)]
net = torch.empty(m, 2 * d, device=dev)
x1, x1b, x1t = torch.empty(m, d, device=dev), torch.empty(m, d, **bf), torch.empty(d, mp, **bf)
ex = torch.empty(m, d, device=dev)
def separate():
    ops.made_chain(xb, m, hidden(True) + [dict(w_packed=packed[L - 1][0], n=2 * d, k=d, bias=bs[L - 1], out_f32=net)])
def update():
    lib.call('gv_iaf_update_fwd_bf16', ptr(z), ptr(net), 2 * d, ptr(x_old), ptr(cc), ptr(x1), ptr(x1b), x1b.stride(0), ptr(x1t), x1t.stride(0), m, d, lib.stream())
print(f'm={m}: chain -> [mu|alpha]        : {timeit(separate):7.1f} us')
print(f'        update kernel            : {timeit(update):7.1f} us')
def fused(debug=0, net_out=True, copies=True, exo=False, hidden_copies=True):
    head = dict(w_packed=packed_iaf[L - 1][0], n=2 * d, k=d, bias=bs[L - 1], iaf=dict(z=z, x_old=x_old, colcount=cc, x_new=x1, debug=debug))
    if net_out: head['out_f32'] = net
    if exo: head['iaf']['ex'] = ex
    if copies: head.update(out_bf16=x1b, out_bf16_t=x1t)
    ops.made_chain(xb, m, hidden(hidden_copies) + [head])
for name, kw in (('fused, all outputs', {}), ('fused, ex instead of [mu|alpha]', dict(net_out=False, exo=True)),
                 ('fused, x_new only', dict(net_out=False, copies=False)),
                 ('fused, x_new + copies', dict(net_out=False, copies=True)), ('fused, x_new + ex', dict(net_out=False, copies=False, exo=True)),
                 ('fused, x_new + [mu|alpha]', dict(net_out=True, copies=False)),
                 ('  no z load', dict(debug=1)), ('  no f32 stores', dict(debug=2)), ('  no transposed stores', dict(debug=4)),
                 ('  no LDS tile', dict(debug=8)), ('  no expf', dict(debug=16)), ('  none of them', dict(debug=31)),
                 ('fused, no hidden copies', dict(hidden_copies=False)), ('fused bare, no hidden copies', dict(hidden_copies=False, debug=31))):
    print(f'        {name:34s}: {timeit(lambda: fused(**kw)):7.1f} us')
print(f'        chain, no hidden copies  : {timeit(lambda: ops.made_chain(xb, m, hidden(False) + [dict(w_packed=packed[L - 1][0], n=2 * d, k=d, bias=bs[L - 1], out_f32=net)])):7.1f} us')
