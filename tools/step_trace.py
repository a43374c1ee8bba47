"""Wall time per block of 25 steps of the bench pipeline (c2, B=8192) from the very first step: shows warm-up transients."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sibrar_amd as S
import bench
dev = 'cuda:0'
torch.set_num_threads(bench.host_cores())
sys.setswitchinterval(1e-3)
ds, net = bench.build(S, dict(bench.C2), dev)
loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
tr = S.Trainer(net, None, None, loss, bench._Conf(dev))
net.train()
ld = S.NegativeSamplingDataLoader(ds, batch_size=8192, shuffle=True, device=dev, prefetch=4, prepare_fn=tr.fused.prepare)
it = bench.epochs(ld)
t0 = time.perf_counter()
for blk in range(20):
    for _ in range(25):
        tr.train_step(*next(it))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f'steps {blk*25:4d}-{blk*25+24:4d}: {(t1 - t0) / 25 * 1e3:7.3f} ms/step  graphs={sum(1 for v in tr.fused._graphs.values() if v is not None)} replays={tr.fused.n_replays}', flush=True)
    t0 = t1
ld.close(); tr.fused.close()
