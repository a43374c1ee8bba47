"""Lab: the c3 (Onion18-shaped) step at the reference's batch 256 — time per step in the captured step against the sum of its
kernels' own times (HIP events around plain launches): how much of a small-batch step is launch floor?   usage: python tools/lab/c3_small.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import sibrar_amd as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = 'cuda:0'
ds = S.SyntheticDataset(5192, 13610, 326_000, item_dense={'audio': 1024}, item_tags={'genres': (853, 5)}, seed=0, n_negative_samples=10,
                        negative_sampling_strategy='uniform_recbole')
torch.manual_seed(42); np.random.seed(42)
net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(bench.C3_MODEL), ds).to(dev)
bpr = S.RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
dt, timings = bench.bench_training(S, ds, net, dev, B, 100, 5, 0, 1, time_kernels=True, loss=bpr)
rows = bench.kernel_table(timings, 100)
print(f'B={B}: {1e3 * dt / 100:.3f} ms per step (captured); {sum(r["launches_per_step"] for r in rows):.0f} launches per step, '
      f'sum of their times {sum(r["ms_per_step"] for r in rows):.3f} ms')
for r in rows[:25]:
    print(f'  {r["entry_point"]:34s} x{r["launches_per_step"]:4.1f}  {r["avg_launch_ms"] * 1e3:7.1f} us  {r["ms_per_step"]:.4f} ms')
