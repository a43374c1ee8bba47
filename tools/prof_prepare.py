"""Times the host side of a training step (FusedTrainStep.prepare) piece by piece on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sibrar_amd as S
import bench

dev = 'cuda:0'
cfg = dict(bench.C2)
ds, net = bench.build(S, cfg, dev)
loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
tr = S.Trainer(net, None, None, loss, bench._Conf(dev))
net.train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(0)
u = torch.from_numpy(rng.integers(0, ds.n_users, size=B))
i = torch.from_numpy(rng.integers(0, ds.n_items, size=(B, 11)))
l = torch.zeros(B, 11, dtype=torch.float64)
f = tr.fused


def t(name, fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f'{name:30s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms')


t('draw', lambda: f.draw(u.shape, i.shape))
d = f.draw(u.shape, i.shape)
t('plan pad', lambda: f.item.plan(d[1], True))
t('plan nopad', lambda: f.item.plan(d[1], False))
t('pin i', lambda: i.pin_memory())
t('pin+to i', lambda: i.pin_memory().to(dev, non_blocking=True))
t('to i (pageable)', lambda: i.to(dev, non_blocking=True))
t('prepare ahead', lambda: f.prepare(u, i, l, ahead=True))
t('prepare inline', lambda: f.prepare(u, i, l, ahead=False))
t('ext i', lambda: torch.cat([i.reshape(-1), i.reshape(-1)[:1]]))
t('np ext i', lambda: torch.from_numpy(np.concatenate([i.reshape(-1).numpy(), i.reshape(-1).numpy()[:1]])))
t('as_tensor', lambda: (torch.as_tensor(u).long(), torch.as_tensor(i).long(), torch.as_tensor(l).double()))
t('lab contiguous', lambda: l.contiguous())
