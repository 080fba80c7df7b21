import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import ops
from tools.microbench import timeit
M = 40943
bf = dict(dtype=torch.bfloat16, device='cuda')
n = k = 200
a = torch.randn(M, k, device='cuda').to(torch.bfloat16)
b = torch.randn(n, k, device='cuda').to(torch.bfloat16)
Mp = (M + 7) // 8 * 8
cb_stack = torch.empty(5 * M, n, **bf)
ct_stack = torch.empty(n, 5 * Mp, **bf)
ct = torch.empty(n, Mp, **bf)
cb = torch.empty(M, n, **bf)
bias = torch.randn(n, device='cuda')
print('compact  :', timeit(lambda: ops.gemm_bf16_nt(a, b, M, n, k, bias=bias, relu=True, c_bf16=cb, c_bf16_t=ct)))
print('stacked T:', timeit(lambda: ops.gemm_bf16_nt(a, b, M, n, k, bias=bias, relu=True, c_bf16=cb_stack[2*M:3*M], c_bf16_t=ct_stack[:, 2*Mp:2*Mp+M])))
# rotate over the 5 slices like the real step (cold outputs)
i = [0]
def rot():
    j = i[0] % 5; i[0] += 1
    ops.gemm_bf16_nt(a, b, M, n, k, bias=bias, relu=True, c_bf16=cb_stack[j*M:(j+1)*M], c_bf16_t=ct_stack[:, j*Mp:j*Mp+M])
print('rotating :', timeit(rot))
