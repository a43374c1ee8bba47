"""Lab: the c1 (ML-1M-shaped) or c3 (Onion18-shaped) training step at a given batch — time per step in the captured step against
the sum of its kernels' own times (HIP events around plain launches), and the entry points by time per step.
usage: python tools/lab/config_kernels.py [c1|c3|c4] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import sibrar_amd as S
CFG = sys.argv[1] if len(sys.argv) > 1 else 'c3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = 'cuda:0'
if CFG == 'c3':
    ds = S.SyntheticDataset(5192, 13610, 326_000, item_dense={'audio': 1024}, item_tags={'genres': (853, 5)}, seed=0, n_negative_samples=10,
                            negative_sampling_strategy='uniform_recbole')
    model = bench.C3_MODEL
elif CFG == 'c4':
    ds = S.SyntheticDataset(1_000_000, 200_000, 20_000_000, item_dense={'text': 768, 'image': 2048}, seed=0, n_negative_samples=10,
                            negative_sampling_strategy='uniform_recbole')
    model = {'shared_common_dim': 256, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
             'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'image'}], 'single_branch_hidden_layers': [256],
                      'preference_hidden_layers': [], 'common_modality_dim': 256}}
else:
    C1 = bench.C1
    ds = S.SyntheticDataset(C1['n_users'], C1['n_items'], C1['nnz'], item_dense={'text': 768}, item_tags={'genres': (18, 3)}, seed=0,
                            n_negative_samples=C1['n_neg'], negative_sampling_strategy='uniform_recbole', holdout_per_user=1, item_popularity=1.0)
    model = bench.C1_MODEL
torch.manual_seed(42); np.random.seed(42)
net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(model), ds).to(dev)
bpr = (S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10) if CFG == 'c4' else
       S.RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10))
dt, timings = bench.bench_training(S, ds, net, dev, B, 100, 5, 0, 1, time_kernels=True, loss=bpr)
rows = bench.kernel_table(timings, 100)
print(f'{CFG} B={B}: {1e3 * dt / 100:.3f} ms per step (captured); {sum(r["launches_per_step"] for r in rows):.0f} launches per step, '
      f'sum of their times {sum(r["ms_per_step"] for r in rows):.3f} ms')
for r in rows[:25]:
    print(f'  {r["entry_point"]:34s} x{r["launches_per_step"]:4.1f}  {r["avg_launch_ms"] * 1e3:7.1f} us  {r["ms_per_step"]:.4f} ms')
