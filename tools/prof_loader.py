"""Producer-side throughput: batches/s of NegativeSamplingDataLoader alone (no training), with and without prepare()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sibrar_amd as S
import bench

dev = 'cuda:0'
ds, net = bench.build(S, dict(bench.C2), dev)
loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
tr = S.Trainer(net, None, None, loss, bench._Conf(dev))
net.train()
for B in (8192, 256):
    for name, kw in (('collate only', {}), ('collate+prepare', {'prepare_fn': tr.fused.prepare})):
        ld = S.NegativeSamplingDataLoader(ds, batch_size=B, shuffle=True, device=dev, prefetch=0, **kw)
        it = iter(ld)
        for _ in range(5):
            next(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            next(it)
        torch.cuda.synchronize()
        print(f'B={B:5d} {name:18s} {(time.perf_counter() - t0) / n * 1e3:7.3f} ms/batch')
import cProfile, pstats
ld = S.NegativeSamplingDataLoader(ds, batch_size=8192, shuffle=True, device=dev, prefetch=0, prepare_fn=tr.fused.prepare)
it = iter(ld)
next(it)
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    next(it)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
