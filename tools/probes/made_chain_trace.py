#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from gcn_vae_amd import ops
dev = torch.device('cuda:0')
n = int(os.environ.get('N', 14541)); d = 200
bf = dict(dtype=torch.bfloat16, device=dev)
widths = [d, 200, 200, 200, 200, 400]; L = len(widths) - 1
ws = [torch.randn(widths[i + 1], widths[i], device=dev) * 0.1 for i in range(L)]
bs = [torch.randn(widths[i + 1], device=dev) for i in range(L)]
pk = [ops.made_pack_weight(w) for w in ws]
x = torch.randn(n, d, device=dev).to(torch.bfloat16)
npad = (n + 7) // 8 * 8
ob = [torch.empty(n, widths[i + 1], **bf) for i in range(L - 1)]
ot = [torch.empty(widths[i + 1], npad, **bf) for i in range(L - 1)]
net = torch.empty(n, 400, device=dev)
trace = torch.zeros(8 * 128, dtype=torch.int64, device=dev)
def fwd():
    ops.made_chain(x, n, [dict(w_packed=pk[l][0], n=widths[l + 1], k=widths[l], bias=bs[l], relu=True, out_bf16=ob[l], out_bf16_t=ot[l]) for l in range(L - 1)] +
                   [dict(w_packed=pk[L - 1][0], n=400, k=200, bias=bs[L - 1], out_f32=net)])
for _ in range(3): fwd()
torch.cuda.synchronize()
os.environ['GV_CHAIN_TRACE'] = str(trace.data_ptr())
fwd(); torch.cuda.synchronize()
t = trace.view(8, 128).cpu()
t0 = min(int(v) >> 4 for v in t[:, 0])
names = {0: 'start', 1: 'staged', 2: 'B0', 3: 'arrive', 4: 'pass', 5: 'stored', 6: 'fence', 7: 'mma', 8: 'epi'}
for w in (0, 6, 7):
    print(f'wave {w}: ' + ' '.join(f'{names[int(v) & 15]}={((int(v) >> 4) - t0) / 100:.2f}' for v in t[w] if int(v) > 0))
