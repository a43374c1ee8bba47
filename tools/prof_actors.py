"""cProfile of the three Python actors of a B=256 step, each run alone: collate, prepare, launch thread (step)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sibrar_amd as S
import bench

dev = 'cuda:0'
ds, net = bench.build(S, dict(bench.C2), dev)
loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
tr = S.Trainer(net, None, None, loss, bench._Conf(dev))
net.train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ld = S.NegativeSamplingDataLoader(ds, batch_size=B, shuffle=True, device=dev, prefetch=0)
it = iter(ld)
raw = [next(it) for _ in range(260)]
f = tr.fused
prepared = [f.prepare(*b) for b in raw[:60]]
for b, pb in zip(raw[:10], prepared[:10]):
    tr.train_step(*b, pb)
torch.cuda.synchronize()


def prof(name, fn, n):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for k in range(n):
        fn(k)
    pr.disable()
    torch.cuda.synchronize()
    print(f'==== {name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call')
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)


prof('collate', lambda k: next(it), 100)
prof('prepare', lambda k: f.prepare(*raw[60 + k]), 100)
prof('step (prepared)', lambda k: tr.train_step(*raw[10 + k], prepared[10 + k]), 50)
