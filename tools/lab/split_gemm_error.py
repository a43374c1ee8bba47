"""Error of the bf16x3-split GEMM and of the fp32-pipe GEMM against an fp64 product (max and rms of |err| / (|x| @ |w|))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sibrar_amd as S
ops = S.ops
dev = 'cuda'
g = torch.Generator().manual_seed(3)
for M, spread in [(4097, 0), (4097, 3), (90112, 0), (90112, 3)]:
    x = torch.randn(M, 128, generator=g)
    if spread:
        x = x * torch.pow(10., torch.randint(-spread, spread, (M, 128), generator=g).float())
    w = torch.randn(128, 128, generator=g) / 8
    x, w = x.to(dev), w.to(dev)
    ref = x.double() @ w.double().t()
    mag = x.double().abs() @ w.double().abs().t()
    for split in (True, False):
        ops._SPLIT, ops._SPLIT_MIN_ROWS = split, 1
        out = ops.linear_nt(x, w, None, 0)
        e = (out.double() - ref).abs() / mag
        print(f'M={M} spread=1e±{spread} {"split" if split else "fp32 "}: max {e.max().item():.3e}  rms {e.pow(2).mean().sqrt().item():.3e}  '
              f'mean signed {((out.double() - ref) / mag).mean().item():+.3e}')
