#!/usr/bin/env python3
"""In-kernel timeline of gv_made_chain_f32 (GV_C32_DEBUG=16): s_memtime stamps of workgroup 0's eight waves -- per unit: entered (after
the layer barrier), fragments landed .. MFMAs done, epilogue done."""
import os, sys
os.environ['GV_C32_DEBUG'] = str(16 | int(os.environ.get('GV_C32_DEBUG', '0')))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import made, ops
from gcn_vae_amd.flows import MADE

d, m = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 14741
torch.manual_seed(0)
mod = MADE(d, d, 3).cuda()
lin = mod._linears()
L = len(lin)
ws = [ops.masked_weight(l.mask, l.weight).detach() for l in lin]
bs = [l.bias.detach() for l in lin]
masks = [l.mask for l in lin]
widths, kin = [w.shape[0] for w in ws], [w.shape[1] for w in ws]
packed = made.made_pack_weights_f32(ws)
x = torch.randn(m, d, device='cuda')
acts = [torch.empty(m, widths[l], device='cuda') for l in range(L)]
pf = made.made_chain_f32_plan(widths, kin, masks)
big = torch.zeros(made.PLAN_WORDS + 8 * 64, dtype=torch.int32, device='cuda')
big[:made.PLAN_WORDS] = pf
fn = lambda: made.made_chain_f32(x, m, [dict(w_packed=packed[l][0], n=widths[l], k=kin[l], bias=bs[l], relu=l < L - 1, out_f32=acts[l]) for l in range(L)], big)
for _ in range(5):
    fn()
torch.cuda.synchronize()
ts = big[made.PLAN_WORDS:].cpu().numpy().astype('int64').reshape(8, 64) & 0xffffffff
plan = big[:made.PLAN_WORDS].cpu().numpy()
t0 = ts[:, 0].min()
for w in range(8):
    lst = plan[4 + (w & 3) * 256: 4 + (w & 3) * 256 + plan[w & 3]][(w >> 2)::2]          # (GV_C32_SPLIT=1 builds: both waves walk the whole list)
    row = ts[w]
    n = int((row != 0).sum())
    rel = [(int(v) - int(t0)) & 0xffffffff for v in row[:n]]
    out = [f'start {rel[0]}']
    for i, e in enumerate(lst):
        b = 1 + 5 * i
        if b + 4 < n:
            sets = plan[4 + 4 * 256 + 2 * ((e >> 8) * 16 + (e & 0x7f))]
            out.append(f'L{e >> 8}t{e & 0xff}: enter {rel[b]} mma {rel[b + 2] - rel[b + 1]} open+issue {rel[b + 3] - rel[b + 2]} epi {rel[b + 4] - rel[b + 3]} end {rel[b + 4]}')
    out.append(f'exit {rel[n - 1]}')
    print(f'wave {w}: ' + ' | '.join(out))
