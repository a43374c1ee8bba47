import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench, sibrar_amd as S
ds, net = bench.build(S, dict(bench.C2), 'cuda:0')
loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
tr = S.Trainer(net, None, None, loss, bench._Conf('cuda:0'))
net.train()
it = iter(S.NegativeSamplingDataLoader(ds, batch_size=8192))
for _ in range(3):
    tr.train_step(*next(it))
torch.cuda.synchronize()
# pure loader time
t0 = time.perf_counter()
batches = [next(it) for _ in range(10)]
print('loader ms/batch', (time.perf_counter() - t0) * 100)
t0 = time.perf_counter()
for b in batches:
    tr.train_step(*b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('step host ms', (t1 - t0) * 100, 'incl sync ms', (t2 - t0) * 100)
pr = cProfile.Profile()
pr.enable()
for b in batches:
    tr.train_step(*b)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
