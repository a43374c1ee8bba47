"""cProfile of one fused evaluation pass on the c2 world (host-side hot spots of evaluate_recommender_algorithm)."""
import os, sys, time, cProfile, pstats
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
L = type('L', (), {'dataset': ev, 'batch_size': 256})()
def run():
    evaluator = S.FullEvaluator(config=S.evaluation._Cfg(top_k=(1, 10, 20)), dataset=ev)
    m = S.evaluate_recommender_algorithm(net, L, evaluator, dev, scorer='fp16_fused')
    torch.cuda.synchronize()
    return m
for _ in range(3): run()
t0 = time.perf_counter(); run(); print('one pass: %.2f ms' % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
