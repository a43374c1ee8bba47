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
rowmap = torch.randperm(R, generator=g).to(torch.int32)
m = torch.nn.Embedding(R, D).to(DEV)
opt = S.FusedOptimizer(m, 'adamw', lr=3e-3, weight_decay=1e-2)
d = engine.DeferredTable(opt, m.weight, 0, R * D, rowmap.to(DEV))
opt.deferred = d
rng = np.random.default_rng(2)
for t in range(15):
    n = int(rng.integers(1, 12))
    ids = torch.from_numpy(rng.integers(0, R if t % 7 else 5, size=n))
    rows = rowmap[ids].long().unique()
    ids_dev = ids.to(DEV)
    d.catch_up(ids_dev)
    m.weight.grad[rows.to(DEV)] = torch.randn(len(rows), D, generator=g).to(DEV)
    opt.step_flat(skip=(0, R * D)); d.update(ids_dev)
    torch.cuda.synchronize()
    nz = (m.weight.grad.abs().sum(1) > 0).nonzero().flatten().tolist()
    print(t, 'ids', ids.tolist(), 'rows', rows.tolist(), 'nonzero grad rows after update', nz, 'last of rows', d.last[rows.to(DEV)].tolist(), 'claim', d.claim[rows.to(DEV)].tolist())
