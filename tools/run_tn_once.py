import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sibrar_amd as S
g = torch.Generator(device='cuda').manual_seed(0)
H = torch.randn(90112, 128, device='cuda', generator=g)
dZ = torch.randn(90112, 128, device='cuda', generator=g)
W2 = torch.randn(128, 128, device='cuda', generator=g)
for _ in range(3):
    S.ops.matmul_tn(dZ, H)
    S.ops.linear_nt(H, W2, None, 1)
torch.cuda.synchronize()
