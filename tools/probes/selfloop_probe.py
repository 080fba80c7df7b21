#!/usr/bin/env python3
"""The R-GCN self-loop products at FB15k-237 size (14 541 x 200 x 200 / x 400) on the generic fp32 GEMM and as a one-layer
gv_made_chain_f32 (dense plan): us per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import made, ops
from tools.microbench import timeit

m = int(sys.argv[1]) if len(sys.argv) > 1 else 14541
for k, n in ((200, 200), (200, 400), (400, 200)):
    x = torch.randn(m, k, device='cuda')
    w = torch.randn(k, n, device='cuda') * 0.05          # loop_weight (in, out): y = x @ w
    b = torch.randn(n, device='cuda')
    out = torch.empty(m, n, device='cuda')
    t_gemm = timeit(lambda: ops.gemm(x, w, bias=b, out=out))
    wt = w.t().contiguous()                                # as a "layer" W (n, k): forward B = W^T = w
    packed = made.made_pack_weights_f32([wt], bwd=False)
    plan = made.made_chain_f32_plan([n], [k], None)
    out2 = torch.empty(m, n, device='cuda')
    t_chain = timeit(lambda: made.made_chain_f32(x, m, [dict(w_packed=packed[0][0], n=n, k=k, bias=b, out_f32=out2)], plan))
    t_pack = timeit(lambda: made.made_pack_weights_f32([wt], bwd=False))
    ops.gemm(x, w, bias=b, out=out)
    same = torch.equal(out, out2)
    print(f'm={m} k={k} n={n}: gv_gemm_f32 {t_gemm:6.1f} us, one-layer chain {t_chain:6.1f} us (+ pack {t_pack:4.1f} us), bit-identical: {same}', flush=True)
