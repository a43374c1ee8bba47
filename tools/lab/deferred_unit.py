"""Debug aid: deferred vs dense Adam on deterministic gradients, difference after every step (flush each step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sibrar_amd as S
from importlib import import_module
engine = import_module(S.ops.__name__.rsplit('.', 1)[0] + '.engine')
DEV = 'cuda'
R, D = 300, 48
g = torch.Generator().manual_seed(5)
w0 = torch.randn(R, D, generator=g) * 0.1
mods, opts = [], []
for _ in range(2):
    m = torch.nn.Embedding(R, D)
    with torch.no_grad(): m.weight.copy_(w0)
    m.to(DEV); mods.append(m); opts.append(S.FusedOptimizer(m, 'adamw', lr=3e-3, weight_decay=1e-2))
d = engine.DeferredTable(opts[1], mods[1].weight, 0, R * D, None)
opts[1].deferred = d
rng = np.random.default_rng(2)
flush_every = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for t in range(12):
    n = int(rng.integers(1, 12))
    ids = torch.from_numpy(rng.integers(0, R, size=n))
    rows = ids.unique()
    grad_rows = torch.randn(len(rows), D, generator=g)
    opts[0].zero_grad(); mods[0].weight.grad[rows.to(DEV)] = grad_rows.to(DEV); opts[0].step_flat()
    ids_dev = ids.to(DEV)
    d.catch_up(ids_dev)
    mods[1].weight.grad[rows.to(DEV)] = grad_rows.to(DEV)
    opts[1].step_flat(skip=(0, R * D)); d.update(ids_dev)
    if (t + 1) % flush_every == 0:
        d.flush()
        a, b = mods[0].weight.detach().cpu(), mods[1].weight.detach().cpu()
        bad = (a != b).any(1)
        print(f't={t+1} touched={sorted(rows.tolist())[:6]}.. max diff {float((a-b).abs().max()):.3e} rows differing {int(bad.sum())} '
              f'(touched among them {int(bad[rows].sum())}) m diff {float((opts[0].m-opts[1].m).abs().max()):.2e} v diff {float((opts[0].v-opts[1].v).abs().max()):.2e}')
