"""Runs the BASELINE configs c1 (ML-1M shape) and c3 (Onion18 shape) end to end on the GPU: training steps through the
Trainer/loader pipeline + one full evaluation. Shapes from SURVEY.md §8(d). Prints step time, interactions/s and NDCG@10."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sibrar_amd as S
import bench

dev = 'cuda:0'
torch.set_num_threads(bench.host_cores())

CFGS = {
    'c1': dict(ds=dict(n_users=5816, n_items=3299, nnz=651034, item_dense={'text': 768}, item_tags={'genres': (18, 3)}),
               model={'shared_common_dim': 64, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
                      'item': {'features': [{'feature_name': 'genres'}, {'feature_name': 'text'}],
                               'single_branch_hidden_layers': [64], 'preference_hidden_layers': [], 'common_modality_dim': 64,
                               'embedding_regularization_type': 'pairwise_single', 'regularization_temperature': 0.1,
                               'regularization_weight': 1e-3, 'normalize_single_branch_input': True}},
               loss='bpr', batch=256),
    'c1-both-entities': dict(ds=dict(n_users=5816, n_items=3299, nnz=651034, item_dense={'text': 768}, item_tags={'genres': (18, 3)},
                                     user_categorical={'gender': 2}),
               model={'shared_common_dim': 64,
                      'user': {'features': [{'feature_name': 'interactions'}, {'feature_name': 'gender'}],
                               'single_branch_hidden_layers': [64], 'preference_hidden_layers': [], 'common_modality_dim': 64,
                               'embedding_regularization_type': 'pairwise_single', 'regularization_temperature': 0.1,
                               'regularization_weight': 1e-3},
                      'item': {'features': [{'feature_name': 'interactions'}, {'feature_name': 'genres'}, {'feature_name': 'text'}],
                               'single_branch_hidden_layers': [64], 'preference_hidden_layers': [], 'common_modality_dim': 64,
                               'embedding_regularization_type': 'pairwise_single', 'regularization_temperature': 0.1,
                               'regularization_weight': 1e-3}},
               loss='bpr', batch=256),
    # the shipped conf/single/algorithms/sbnet_ml1m_conf.yml: both sides entities, CSR interactions modality on both, input
    # dropout 0.2 on the item side, no embedding regularisation, BPR, AdamW, batch 256
    'ml1m-shipped': dict(ds=dict(n_users=5816, n_items=3299, nnz=651034, item_dense={'plot_mpnet': 768}, item_tags={'genres': (18, 3)},
                                 user_categorical={'gender': 2, 'occupation': 21}),
               model={'shared_common_dim': 64,
                      'user': {'features': [{'feature_name': 'interactions', 'feature_hidden_layers': []},
                                            {'feature_name': 'gender', 'feature_hidden_layers': []},
                                            {'feature_name': 'occupation', 'feature_hidden_layers': []}],
                               'single_branch_hidden_layers': [], 'preference_hidden_layers': [], 'common_modality_dim': 64,
                               'activation_fn': 'relu', 'single_branch_input_dropout': None},
                      'item': {'features': [{'feature_name': 'interactions', 'feature_hidden_layers': []},
                                            {'feature_name': 'genres', 'feature_hidden_layers': []},
                                            {'feature_name': 'plot_mpnet', 'feature_hidden_layers': []}],
                               'single_branch_hidden_layers': [64], 'preference_hidden_layers': [], 'common_modality_dim': 64,
                               'activation_fn': 'relu', 'single_branch_input_dropout': 0.2}},
               loss='bpr', batch=256),
    # BASELINE c4 shape on ONE GPU (the 8-GPU run shards the batch): 1M users x 200k items, text 768 + image 2048, D = 256,
    # user = embedding lookup; 20M interactions keep the host-side generation short
    'c4': dict(ds=dict(n_users=1_000_000, n_items=200_000, nnz=20_000_000, item_dense={'text': 768, 'image': 2048}),
               model={'shared_common_dim': 256, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
                      'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'image'}],
                               'single_branch_hidden_layers': [256], 'preference_hidden_layers': [], 'common_modality_dim': 256}},
               loss='ssm', batch=256),
    'c3': dict(ds=dict(n_users=5192, n_items=13610, nnz=326000, item_dense={'audio': 1024}, item_tags={'genres': (853, 5)}),
               model={'shared_common_dim': 128, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
                      'item': {'features': [{'feature_name': 'interactions'}, {'feature_name': 'genres'}, {'feature_name': 'audio'}],
                               'single_branch_hidden_layers': [512, 512, 512, 256, 256], 'preference_hidden_layers': [],
                               'common_modality_dim': 512, 'embedding_regularization_type': 'pairwise_single',
                               'regularization_temperature': 0.1, 'regularization_weight': 1e-4}},
               loss='bpr', batch=256),
}
args = [a for a in sys.argv[1:] if not a.startswith('--')]
SWEEPS = [int(a.split('=')[1]) for a in sys.argv if a.startswith('--sweep=')]      # lab: sweep period of the deferred table (0: none)
if SWEEPS:
    from importlib import import_module
    import_module(S.ops.__name__.rsplit('.', 1)[0] + '.engine').DeferredTable.SWEEP_EVERY = SWEEPS[0]
TRAIN_ONLY = '--train-only' in sys.argv
BATCHES = [int(a.split('=')[1]) for a in sys.argv if a.startswith('--batch=')]
which = args or list(CFGS)
for name in which:
    c = CFGS[name]
    ds = S.SyntheticDataset(seed=0, n_negative_samples=10, negative_sampling_strategy='uniform_recbole', holdout_per_user=2,
                            **c['ds'])
    torch.manual_seed(42); np.random.seed(42)
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(c['model']), ds).to(dev)
    n_par = sum(p.numel() for p in net.parameters())
    loss = (S.RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
            if c['loss'] == 'bpr' else
            S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10))
    tr = S.Trainer(net, None, None, loss, bench._Conf(dev))
    net.train()
    for B in (BATCHES or (c['batch'], 4096)):
        ld = S.NegativeSamplingDataLoader(ds, batch_size=B, shuffle=True, device=dev, prefetch=4,
                                          prepare_fn=tr.fused.prepare if tr.fused is not None else None)
        it = bench.epochs(ld)
        for _ in range(25):
            out = tr.train_step(*next(it))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 60
        for _ in range(n):
            out = tr.train_step(*next(it))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ld.close()
        # the three actors alone: collate, prepare, launch thread on prepared batches (GPU-bound when the host keeps up)
        ld0 = S.NegativeSamplingDataLoader(ds, batch_size=B, shuffle=True, device=dev, prefetch=0)
        it0 = iter(ld0)
        t0 = time.perf_counter(); raw = [next(it0) for _ in range(40)]; t_col = (time.perf_counter() - t0) / 40
        t0 = time.perf_counter(); prep = [tr.fused.prepare(*b) for b in raw]; t_prep = (time.perf_counter() - t0) / 40
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b, pb in zip(raw, prep):
            tr.train_step(*b, pb)
        torch.cuda.synchronize()
        t_gpu = (time.perf_counter() - t0) / 40
        print(f'{name:18s}   alone: collate {t_col*1e3:.3f}  prepare {t_prep*1e3:.3f}  step-on-prepared {t_gpu*1e3:.3f} ms', flush=True)
        print(f'{name:18s} params {n_par/1e6:6.2f}M  B={B:5d}  {dt*1e3:7.3f} ms/step  {B/dt/1e3:9.1f} k interactions/s  '
              f'loss {float(out[0]):.4f}  fused={tr.fused is not None} replays={tr.fused.n_replays if tr.fused else 0}', flush=True)
    if TRAIN_ONLY:
        tr.fused.close()
        continue
    net.eval()
    ev = ds.eval_view()
    t0 = time.perf_counter()
    for scorer in ('fp32', 'fp16_fused'):
        evaluator = S.FullEvaluator(config=S.evaluation._Cfg(top_k=(1, 10, 20)), dataset=ev)
        m = S.evaluate_recommender_algorithm(net, type('L', (), {'dataset': ev, 'batch_size': 2048})(), evaluator, dev,
                                             scorer=scorer)
        torch.cuda.synchronize()
        print(f'{name:18s} eval[{scorer}] ndcg@10 {m.get("ndcg@10", float("nan")):.5f} recall@10 {m.get("recall@10", float("nan")):.5f}  '
              f'{(time.perf_counter() - t0)*1e3:.1f} ms', flush=True)
        t0 = time.perf_counter()
    if tr.fused:
        tr.fused.close()
