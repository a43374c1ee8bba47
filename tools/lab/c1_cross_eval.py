"""Lab: where does the NDCG@10 difference of bench.py's c1 object come from — the training trajectories or the evaluation? Trains the
CPU port and the GPU engine on the same 300 recorded batches (as bench_c1 does), then evaluates BOTH parameter sets with BOTH
evaluators (CPU: oracle/eval_ref; GPU: evaluate_recommender_algorithm, fp32 scorer).   usage: python tools/lab/c1_cross_eval.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import sibrar_amd as S
from oracle import eval_ref, losses_ref, model_ref, sampling_ref, train_ref
dev = 'cuda:0'
C1, C1_MODEL = bench.C1, bench.C1_MODEL
ds = S.SyntheticDataset(C1['n_users'], C1['n_items'], C1['nnz'], item_dense={'text': 768}, item_tags={'genres': (18, 3)}, seed=0,
                        n_negative_samples=C1['n_neg'], negative_sampling_strategy='uniform_recbole', holdout_per_user=1, item_popularity=1.0)
torch.manual_seed(42); np.random.seed(42)
net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(C1_MODEL), ds).to(dev)
sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
bpr = S.RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=ds.n_negative_samples)
sd = {k: v.clone() for k, v in sd0.items()}
for v in sd.values():
    if v.dtype.is_floating_point:
        v.requires_grad_(True)
ut = {'user_embedding': model_ref.RefTable('categorical', np.arange(ds.n_users), n_categories=ds.n_users)}
it = {k: model_ref.table_from_feature(f) for k, f in ds.item_features.items()}
orders = {'item_train': net.item_embedding_module.train_modality_order, 'item_eval': net.item_embedding_module.eval_modality_order}
ref = model_ref.RefSingleBranchNet(sd, C1_MODEL, ut, it, orders=orders)
rloss = losses_ref.RefRecLoss('bpr', n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=ds.n_negative_samples)
opt = train_ref.make_optimizer('adamw', [p for k, p in sd.items() if p.requires_grad and 'running' not in k], 1e-3, 1e-6)
inter = ds.user_sampling_matrix
positives = [inter.indices[inter.indptr[u]:inter.indptr[u + 1]] for u in range(ds.n_users)]
coo = ds.interaction_matrix
rng = np.random.default_rng(0); np.random.seed(42)
recorded = []
for s_ in range(bench.C1_TRAIN_STEPS):
    sel = rng.integers(0, coo.nnz, size=256)
    u, i, l = sampling_ref.recbole_collate(coo.row[sel], coo.col[sel], ds.n_negative_samples, ds.items_in_split, positives)
    mods = ref.sides['item'].sample_modalities(i.shape, True)
    train_ref.train_step(ref, rloss, opt, torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(l), None, mods)
    recorded.append((u, i, l, mods))
ev = ds.eval_view()
excl, labels = ev.exclude_data.tocsr(), ev.user_sampling_matrix.tocsr()


def cpu_eval():
    with torch.no_grad():
        i_repr = ref.item_repr(torch.arange(ds.n_items), False)
        nd = []
        for lo in range(0, ds.n_users, 256):
            ub = torch.arange(lo, min(lo + 256, ds.n_users))
            nd.append(eval_ref.evaluate(ref.user_repr(ub, False), i_repr, excl[lo:lo + 256].toarray(), labels[lo:lo + 256].toarray(), ks=(10,))['ndcg@10'])
    return float(torch.cat(nd).mean())


def gpu_eval():
    evaluator = S.FullEvaluator(config=S.evaluation._Cfg(top_k=(10,), metrics=['ndcg'], calculate_std=False), dataset=ev)
    return S.evaluate_recommender_algorithm(net, type('L', (), {'dataset': ev, 'batch_size': 256})(), evaluator, dev, scorer='fp32')['ndcg@10']


cpu_sd = {k: v.detach().clone() for k, v in sd.items()}
res = {'cpu params / cpu eval': cpu_eval()}
net.train()
gopt = S.FusedOptimizer(net, 'adamw', lr=1e-3, weight_decay=1e-6)
fused = S.FusedTrainStep(net, bpr, gopt)
order = list(net.item_embedding_module.train_modality_order)
lut = {m: q for q, m in enumerate(order)}
for u, i, l, mods in recorded:
    pos = np.vectorize(lut.__getitem__, otypes=[np.int8])(mods).reshape(-1, mods.shape[-1])
    fused.step(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(l), (None, (pos, order)))
fused.close()
torch.cuda.synchronize()
gpu_sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
res['gpu params / gpu eval'] = gpu_eval()
with torch.no_grad():
    for k, v in sd.items():
        v.copy_(gpu_sd[k])
res['gpu params / cpu eval'] = cpu_eval()
net.load_state_dict({k: v.to(dev) for k, v in cpu_sd.items()})
res['cpu params / gpu eval'] = gpu_eval()
for k, v in res.items():
    print(f'{k}: NDCG@10 {v:.6f}')
worst = max((float((cpu_sd[k].double() - gpu_sd[k].double()).norm() / (cpu_sd[k].double().norm() + 1e-30)), k) for k in cpu_sd if cpu_sd[k].dtype.is_floating_point)
print('largest relative parameter difference after training:', worst)
