"""fp32 evaluation path (GEMM + CSR mask + exact top-k) on c2 for different user chunk sizes, top-100."""
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
net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(bench.model_config(cfg['emb_dim'])), ds).to(dev).eval()
ev = ds.eval_view()
for chunk in (512, 1024, 2048, 4096, 8192, 16384):
    for rep in range(2):
        evaluator = S.FullEvaluator(config=S.evaluation._Cfg(top_k=(1, 10, 100)), dataset=ev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = S.evaluate_recommender_algorithm(net, type('L', (), {'dataset': ev, 'batch_size': 256})(), evaluator, dev, scorer='fp32',
                                             user_chunk=chunk)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'fp32 top-100, user chunk {chunk:6d}: {dt*1e3:8.1f} ms  ndcg@10 {m["ndcg@10"]:.5f}', flush=True)
