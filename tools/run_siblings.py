"""Sibling models of SURVEY 8(f).4 on the ML-1M-shaped synthetic world (5,816 users x 3,299 items, 651k interactions, 768-d text +
genre tags), the reference's default batch 256 and 4096: ms per training step through Trainer / loader on the GPU (autograd over
the HIP kernels + fused optimizer), one full evaluation, and the CPU oracle's time for the same step on this box's host cores."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sibrar_amd as S
import bench
from oracle import model_ref, losses_ref, train_ref

dev = 'cuda:0'
cores = bench.host_cores()
torch.set_num_threads(cores)
ds = S.SyntheticDataset(5816, 3299, 651034, item_dense={'text': 768}, item_tags={'genres': (18, 3)}, seed=0, n_negative_samples=10,
                        negative_sampling_strategy='uniform_recbole', holdout_per_user=2)
common = dict(aggregate_for_rec=False, lambda_content=1e-4, temperature=0.1, embedding_loss_aggregator='mean', intermediate_layers=[128],
              embedding_dim=64, use_user_bias=False, use_item_bias=True, use_global_bias=True)
CONFS = {
    'mf': dict(embedding_dim=64, use_user_bias=False, use_item_bias=True, use_global_bias=True),
    'ifeatmf': dict(feature_name='text', **common),
    'dropoutnet': dict(user=dict(features=[], preference_layers=[128], common_hidden_layers=[128]),
                       item=dict(features=[dict(feature_name='text', embedding_dim=64), dict(feature_name='genres', embedding_dim=16)],
                                 preference_layers=[128], common_hidden_layers=[128]), shared_common_dim=64),
}
which = [a for a in sys.argv[1:] if not a.startswith('--')] or list(CONFS)
inter = ds.user_sampling_matrix_train
inter_t = ds.item_sampling_matrix_train
for name in which:
    torch.manual_seed(42); np.random.seed(42)
    net = S.ALGORITHMS[name].build_from_conf(CONFS[name], ds).to(dev)
    loss = S.RecBinaryCrossEntropy(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
    conf = bench._Conf(dev)
    conf.learn['lr'] = 1e-3
    tr = S.Trainer(net, None, None, loss, conf)
    net.train()
    n_par = sum(p.numel() for p in net.parameters())
    for B in (256, 4096):
        ld = S.NegativeSamplingDataLoader(ds, batch_size=B, shuffle=True, device=dev, prefetch=4)
        it = bench.epochs(ld)
        for _ in range(15):
            out = tr.train_step(*next(it))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        for _ in range(n):
            out = tr.train_step(*next(it))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ld.close()
        # CPU oracle: the same step (same parameters at this point, torch-CPU fp32, torch.optim.AdamW)
        sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in net.state_dict().items()}
        opt = train_ref.make_optimizer('adamw', [p for p in sd.values() if p.requires_grad], 1e-3, 1e-6)
        ref_loss = losses_ref.RefRecLoss('bce', n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
        it_t = {'text': model_ref.table_from_feature(ds.item_features['text']), 'genres': model_ref.table_from_feature(ds.item_features['genres'])}
        u, i, lab = next(iter(S.NegativeSamplingDataLoader(ds, batch_size=B, shuffle=True)))
        rng = np.random.default_rng(0)

        def cpu_step():
            opt.zero_grad()
            if name == 'mf':
                logits, reg = model_ref.mf_logits(sd, u, i), 0.
            elif name == 'ifeatmf':
                logits, reg = model_ref.feature_mf_forward(sd, 'item', it_t['text'], u, i, embedding_dim=64, intermediate_layers=[128],
                                                            aggregate_for_rec=False, temperature=0.1)
            else:
                cfg = dict(CONFS['dropoutnet'])
                logits = model_ref.dropoutnet_forward(sd, cfg, {}, it_t, inter, inter_t, u, i, rng.choice([1, 2], size=len(u)),
                                                      rng.choice([1, 2], size=len(i)), training=True)
                reg = 0.
            (ref_loss.compute_loss(logits, lab) + reg).backward()
            opt.step()
        cpu_step()
        t0 = time.perf_counter()
        reps = 3 if name == 'dropoutnet' and B == 4096 else 8
        for _ in range(reps):
            cpu_step()
        t_cpu = (time.perf_counter() - t0) / reps
        print(f'{name:11s} params {n_par/1e6:5.2f}M  B={B:5d}  GPU {dt*1e3:7.3f} ms/step ({B/dt/1e3:8.1f} k interactions/s)  '
              f'CPU oracle {t_cpu*1e3:8.1f} ms/step on {cores} cores ({t_cpu/dt:6.0f}x)  loss {float(out[0]):.4f}', flush=True)
    net.eval()
    ev = ds.eval_view()
    evaluator = S.FullEvaluator(config=S.evaluation._Cfg(top_k=(1, 10, 20)), dataset=ev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = S.evaluate_recommender_algorithm(net, type('L', (), {'dataset': ev, 'batch_size': 2048})(), evaluator, dev)
    torch.cuda.synchronize()
    print(f'{name:11s} full evaluation {1e3*(time.perf_counter()-t0):.1f} ms  ndcg@10 {m["ndcg@10"]:.5f}', flush=True)
