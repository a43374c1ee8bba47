"""Line-level timing of FusedTrainStep.prepare at B=8192 (c2) on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sibrar_amd as S
import bench
dev = 'cuda:0'
torch.set_num_threads(bench.host_cores())
ds, net = bench.build(S, dict(bench.C2), dev)
loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
tr = S.Trainer(net, None, None, loss, bench._Conf(dev))
net.train()
f = tr.fused
B = 8192
rng = np.random.default_rng(0)
u = torch.from_numpy(rng.integers(0, ds.n_users, size=B)); i = torch.from_numpy(rng.integers(0, ds.n_items, size=(B, 11)))
l = torch.zeros(B, 11, dtype=torch.float64)
import cProfile, pstats
for _ in range(5):
    f.prepare(u, i, l, labels_key='k')
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(50):
    f.prepare(u, i, l, labels_key='k')
dt = (time.perf_counter() - t0) / 50
pr.disable()
print(f'prepare: {dt*1e3:.3f} ms')
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
