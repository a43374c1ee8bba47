"""Lab: host-side timeline of the bench's timed region (c2, B = 8192): when each of the K steps has been SUBMITTED (launch thread
returns from train_step) relative to the start, and when the GPU has finished all of them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import sibrar_amd as S
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
ds, net = bench.build(S, bench.C2, dev)
loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=ds.n_negative_samples)
trainer = S.Trainer(net, None, None, loss, bench._Conf(dev))
net.train()
sys.setswitchinterval(1e-3)
np.random.seed(42)
loader = S.NegativeSamplingDataLoader(ds, batch_size=8192, shuffle=True, rank=0, world=1, device=dev, dp_sampling='local', prefetch=4,
                                      prepare_fn=trainer.fused.prepare)
it = bench.epochs(loader)
bench.run_steps(S, trainer, it, 40, 1)
import gc
gc.collect(); gc.disable()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for _ in range(K):
        trainer.train_step(*next(it))
        marks.append(time.perf_counter() - t0)
    t_sub = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    d = np.diff([0.0] + marks) * 1e3
    print(f'K={K}: all submitted after {t_sub * 1e3:.2f} ms, GPU done after {t_all * 1e3:.2f} ms = {t_all / K * 1e3:.3f} ms per step; '
          f'submit gaps (ms): first {d[0]:.3f}, median {np.median(d):.3f}, max {d.max():.3f}', flush=True)
loader.close()
trainer.fused.close()
