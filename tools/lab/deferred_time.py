"""Deferred row-wise AdamW against the dense launch at the c2 shape: a [100k, 128] lookup table of which a step touches B random
rows + 6.5 M other parameters that stay on the dense kernel. Times (HIP events, us per step in the steady state): dense launch
over everything; deferred = catch-up of the batch's rows + dense launch over the rest + update of the batch's rows.
usage: python tools/lab/deferred_time.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sibrar_amd as S
from importlib import import_module
engine = import_module(S.ops.__name__.rsplit('.', 1)[0] + '.engine')
DEV = 'cuda'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
R, D, REST = int(os.environ.get('DT_R', 100_000)), int(os.environ.get('DT_D', 128)), int(os.environ.get('DT_REST', 6_500_000))


class M(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.table = torch.nn.Parameter(torch.randn(R, D) * 0.01)
        self.rest = torch.nn.Parameter(torch.randn(REST) * 0.01)


def ev():
    return torch.cuda.Event(enable_timing=True)


rng = np.random.default_rng(0)
res = {}
for mode in ('dense', 'deferred'):
    m = M().to(DEV)
    opt = S.FusedOptimizer(m, 'adamw', lr=1e-3, weight_decay=1e-6)
    lo = opt.fp.offsets[0]
    hi = lo + R * D
    d = None
    if mode == 'deferred':
        d = engine.DeferredTable(opt, m.table, lo, hi, None)
        opt.deferred = d
    ts = {k: [] for k in ('catch', 'dense', 'update')}
    steps = 60
    for t in range(steps):
        ids = torch.from_numpy(rng.integers(0, R, size=B)).to(DEV)
        m.table.grad[ids] = 0.01                       # rows with gradient (duplicates: the same row)
        m.rest.grad.fill_(0.01)
        e = [ev() for _ in range(4)]
        e[0].record()
        if d is not None:
            d.catch_up(ids)
        e[1].record()
        opt.step_flat(zero_grad=True, rows=ids if d is not None else None)
        e[2].record()
        e[3].record()
        torch.cuda.synchronize()
        if t >= 30:
            ts['catch'].append(e[0].elapsed_time(e[1])); ts['dense'].append(e[1].elapsed_time(e[2])); ts['update'].append(e[2].elapsed_time(e[3]))
    res[mode] = {k: 1e3 * sum(v) / len(v) for k, v in ts.items()}
    print(mode, {k: round(v, 1) for k, v in res[mode].items()}, 'us; total', round(sum(res[mode].values()), 1), flush=True)
