"""Wide bf16-split GEMM (csrc/gemm_split_wide_f32.hip) against the fp32 ring kernel on the c3 / c4 shapes: median launch time (HIP
events) of ops.linear_nt / ops.matmul_nn with ops._SPLIT on and off.   usage: python tools/lab/wide_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sibrar_amd as S
ops = S.ops
dev = 'cuda:0'


def t_us(fn, warm=4, reps=10):
    for _ in range(warm): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return 1e3 * ts[len(ts) // 2]


g = torch.Generator().manual_seed(0)
for (M, N, K) in ((90112, 512, 512), (90112, 256, 512), (90112, 512, 256), (90112, 256, 256), (30805, 512, 1024), (5632, 512, 512), (22528, 512, 512), (45056, 256, 256)):
    x = torch.randn(M, K, generator=g).to(dev)
    w_nt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    w_nn = (torch.randn(K, N, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev)
    row = []
    for split in (True, False):
        ops._SPLIT = split
        old = ops._SPLIT_MIN_ROWS
        ops._SPLIT_MIN_ROWS = 1
        ops._WIDE_HEURISTIC = False
        row.append(t_us(lambda: ops.linear_nt(x, w_nt, b, 1, out=out)))
        row.append(t_us(lambda: ops.matmul_nn(x, w_nn, out=out)))
        ops._SPLIT_MIN_ROWS = old
    ops._SPLIT = True
    fl = 2.0 * M * N * K
    print(f'{M} x {N} x {K}: NT wide {row[0]:.0f} us ({fl / row[0] / 1e6:.0f} TF fp32-eq) ring {row[2]:.0f} us ({fl / row[2] / 1e6:.0f} TF)   '
          f'NN wide {row[1]:.0f} us ring {row[3]:.0f} us', flush=True)
