#!/usr/bin/env python3
"""How the forward MADE chain (production form: sign bits, exp(alpha + mu), bf16 operands of the next pass) and the weight-gradient
product scale with the number of rows: us per launch at row counts around the 512 workgroup slots of the chip."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import ops, lib
from tools.microbench import timeit

d, L, S = 200, 5, 5
dev = torch.device('cuda:0')
widths = [d] * L + [2 * d]
ws = [torch.randn(widths[i + 1], widths[i], device=dev) / widths[i] ** 0.5 for i in range(L)]
bs = [torch.randn(widths[i + 1], device=dev) * 0.1 for i in range(L)]
cc = torch.ones(d, dtype=torch.int32, device=dev); cc[-1] = 0
bf = dict(dtype=torch.bfloat16, device=dev)
packed = ops.made_pack_weights(ws, iaf_last=True)
for m in [int(a) for a in sys.argv[1:]] or [16384, 32768, 36864, 40943, 49152, 65536, 98304]:
    x_old = torch.randn(m, d, device=dev); xb = x_old.to(torch.bfloat16); z = torch.randn(m, d, device=dev)
    sign = [torch.empty(m, (d + 31) // 32, dtype=torch.int32, device=dev) for _ in range(L - 1)]
    x1, x1b, ex = torch.empty(m, d, device=dev), torch.empty(m, d, **bf), torch.empty(m, d, device=dev)
    np8 = (m + 7) // 8 * 8
    plain = torch.zeros(d * L, S * np8, **bf)                     # [column of any layer][pass][row]
    p = 2                                                        # the pass whose slice is written
    t_of = lambda l: dict(out_bf16_t=plain[l * d:(l + 1) * d, p * np8:p * np8 + m])
    def chain(copies=True):
        head = dict(w_packed=packed[L - 1][0], n=2 * d, k=d, bias=bs[L - 1], iaf=dict(z=z, x_old=x_old, colcount=cc, x_new=x1, ex=ex, keep=cc),
                    out_bf16=x1b, **(t_of(0) if copies else {}))
        ops.made_chain(xb, m, [dict(w_packed=packed[i][0], n=d, k=d, bias=bs[i], relu=True, out_bits=sign[i], **(t_of(i + 1) if copies else {}))
                               for i in range(L - 1)] + [head])
    gw = torch.zeros(d, d, device=dev); gb = torch.zeros(d, device=dev)
    k8 = S * np8
    split = max(2, min(ops.GRADW_SPLIT_MAX, k8 // 512))
    t1, t2 = timeit(chain), timeit(lambda: chain(False))
    t3 = timeit(lambda: ops.gemm_bf16_gradw(plain[d:2 * d], plain[:d], d, d, k8, gw, a_rowsum=gb, split_k=split))
    print(f'm={m:6d} ({(m + 63) // 64:4d} tiles): forward chain {t1:7.1f} us, without transposed copies {t2:7.1f} us; gradw over {k8} ({split} splits) {t3:7.1f} us', flush=True)
