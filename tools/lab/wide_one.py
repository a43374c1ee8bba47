"""One shape of the wide bf16-split kernel, a few launches (for rocprofv3 --pmc passes).  usage: python tools/lab/wide_one.py M N K mode"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sibrar_amd as S
ops = S.ops
M, N, K, mode = (int(a) for a in sys.argv[1:5])
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).cuda()
w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda() if mode == 0 else (torch.randn(K, N, generator=g) / K ** 0.5).cuda()
out = torch.empty(M, N, device='cuda')
for _ in range(6):
    (ops.linear_nt(x, w, None, 0, out=out) if mode == 0 else ops.matmul_nn(x, w, out=out))
torch.cuda.synchronize()
