"""End-to-end evaluation time (evaluate_recommender_algorithm) on the c2 world: item representations + per-batch user
representations + scoring + metrics, for both scorers and two user batch sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sibrar_amd as S
import bench
dev = 'cuda:0'
torch.set_num_threads(bench.host_cores())
cfg = dict(bench.C2)
ds = S.SyntheticDataset(cfg['n_users'], cfg['n_items'], cfg['nnz'], item_dense={'text': cfg['feat_dim']}, seed=0,
                        n_negative_samples=10, holdout_per_user=2)
torch.manual_seed(42); np.random.seed(42)
net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(bench.model_config(cfg['emb_dim'])), ds).to(dev)
net.eval()
ev = ds.eval_view()
for scorer in ('fp32', 'fp16_fused'):
    for bs in (256, 8192):
        for rep in range(2):
            evaluator = S.FullEvaluator(config=S.evaluation._Cfg(top_k=(1, 10, 20)), dataset=ev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m = S.evaluate_recommender_algorithm(net, type('L', (), {'dataset': ev, 'batch_size': bs})(), evaluator, dev, scorer=scorer)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'{scorer:11s} user batch {bs:5d}: {dt*1e3:8.1f} ms  ({ds.n_users * ds.n_items / dt / 1e9:7.1f} G scores/s)  ndcg@10 {m["ndcg@10"]:.5f}', flush=True)
