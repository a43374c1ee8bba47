"""Timing-only ablations of the fused scorer (SBR_ST_DEBUG: 1 MFMA loop only, 2 + threshold compares, 5/6/7 parts of the MFMA loop) next
to the complete kernel, c2 shape. Results of the ablation builds are meaningless.   usage: python tools/lab/scorer_ablate.py [pre]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sibrar_amd as S
ops = S.ops
dev = 'cuda:0'
U, I, D = 100_000, 50_000, int(os.environ.get('LAB_D', '128'))
if D == 256: I = 25_000
g = torch.Generator().manual_seed(1)
u = (torch.randn(U, D, generator=g) / 8).half().to(dev)
it = (torch.randn(I, D, generator=g) / 8).half().to(dev)
if len(sys.argv) > 1: os.environ['SBR_ST_PRE'] = sys.argv[1]


def t_ms(fn, warm=6, reps=12):
    for _ in range(warm): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2]


for dbg in ('0', '1', '2', '7', '6', '5'):
    os.environ['SBR_ST_DEBUG'] = dbg
    t = t_ms(lambda: ops.score_topk_f16(u, it, 20))
    print(f'D={D} SBR_ST_DEBUG={dbg}: {t:.3f} ms', flush=True)
os.environ['SBR_ST_DEBUG'] = '0'
