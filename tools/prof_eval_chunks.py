"""End-to-end evaluation time of the exact fp32 path on the c2 world for different user chunk sizes (the [chunk, items] fp32 score
matrix is materialised: 16384 users x 50k items = 3.3 GB, 100k users = 20 GB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sibrar_amd as S
import bench
dev = 'cuda:0'
cfg = dict(bench.C2)
ds = S.SyntheticDataset(cfg['n_users'], cfg['n_items'], cfg['nnz'], item_dense={'text': cfg['feat_dim']}, seed=0,
                        n_negative_samples=10, holdout_per_user=2)
torch.manual_seed(42); np.random.seed(42)
net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(bench.model_config(cfg['emb_dim'])), ds).to(dev)
net.eval()
ev = ds.eval_view()
for scorer, chunks in (('fp32', (8192, 16384, 32768, 65536, 100000)), ('fp16_fused', (100000,))):
    for chunk in chunks:
        for rep in range(3):
            evaluator = S.FullEvaluator(config=S.evaluation._Cfg(top_k=(1, 10, 20)), dataset=ev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m = S.evaluate_recommender_algorithm(net, type('L', (), {'dataset': ev, 'batch_size': 256})(), evaluator, dev, scorer=scorer,
                                                 user_chunk=chunk)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'{scorer:11s} user chunk {chunk:6d}: {dt*1e3:8.1f} ms  ndcg@10 {m["ndcg@10"]:.5f}', flush=True)
