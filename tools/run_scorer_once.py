import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sibrar_amd as S
g = torch.Generator(device='cuda').manual_seed(1)
Bu, I, D = 100000, 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 128
u = (torch.randn(Bu, D, device='cuda', generator=g) / 8).half()
it = (torch.randn(I, D, device='cuda', generator=g) / 8).half()
for _ in range(2):
    S.ops.score_topk_f16(u, it, 20)
torch.cuda.synchronize()
