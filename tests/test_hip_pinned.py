"""GPU: the paths that bench.py actually times, pinned directly ("What's weak" 1-4):

  a) the GPU membership kernel of the negative-sampling collate (``sbr_csr_contains`` behind ``DevicePositiveIndex``) against the
     host index, the literal ``v in positives`` test and the reference's golden batch streams (data/dataloader.py:180-191);
  b) ``engine.FusedTrainStep`` — the launch sequence ``value`` is measured on — against the golden groups G4 (losses, every
     gradient, BatchNorm statistics) and G8 (3-step optimizer trajectories) themselves, plain launches, capture and replay;
  c) ``evaluate_recommender_algorithm(scorer='fp16_fused')`` end to end: golden world, a 20k-user synthetic world, the
     fall-backs (k > 32, D outside {64, 128, 256}) and the user chunking;
  d) BASELINE configs c3 / c4 / c5 at their own shapes: one c3 step against the CPU oracle, one c4 step (1M x 200k, D = 256)
     against the module path, the c5 shard shape (100k x 25k x 256, exclusions) against the fp32 GEMM + exact top-k.
Integer work is compared exactly; floating point within the tolerances written at each comparison (north star: 1e-4 relative).
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from golden_util import MANIFEST, I, U, bn_shadowed_biases, close, gscale, load, product_net, state_dict, sub, world

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def S():
    import sibrar_amd
    return sibrar_amd


# ---- a) membership kernel + collate on the GPU --------------------------------------------------------------------------------
def _csr_with_gaps(n_users, n_items, nnz, seed):
    """Random interaction matrix with empty rows, full-range columns (first and last item present) and one long row."""
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, n_users, size=nnz)
    rows[rows % 7 == 3] = (rows[rows % 7 == 3] + 1) % n_users            # users = 3 (mod 7) end up (nearly) empty
    cols = rng.integers(0, n_items, size=nnz)
    long_row = n_users // 2
    extra = np.arange(0, n_items, 2)
    rows = np.concatenate([rows, np.full(len(extra), long_row), [0, 0]])
    cols = np.concatenate([cols, extra, [0, n_items - 1]])
    m = sp.csr_matrix((np.ones(len(rows), dtype=np.int8), (rows, cols)), shape=(n_users, n_items))
    m.sum_duplicates()
    m.data[:] = 1
    m.sort_indices()
    return m


@pytest.mark.parametrize('n_users,n_items,nnz,n_query', [(50, 40, 300, 700), (3000, 2000, 60_000, 50_000),
                                                         (100_000, 50_000, 5_000_000, 90_112)])
def test_csr_contains_kernel_equals_host_membership(n_users, n_items, nnz, n_query):
    """``DevicePositiveIndex.contains`` with the host shortcut off (every query goes through ``sbr_csr_contains``) ==
    ``PositiveIndex.contains`` == the reference's literal ``item in positives_of_user`` on (user, item) pairs of which half are
    true positives; empty rows, row ends, the first and the last item included. The last size is the bench's: the c2 interaction
    matrix and one B = 8192 batch of 81,920 + 8,192 slots."""
    m = _csr_with_gaps(n_users, n_items, nnz, seed=n_users)
    dpi = S().sampling.DevicePositiveIndex(m, DEV)
    dpi.HOST_BELOW = 0
    host = S().sampling.PositiveIndex(m)
    rng = np.random.default_rng(1)
    coo = m.tocoo()
    sel = rng.integers(0, coo.nnz, size=n_query // 2)
    users = np.concatenate([coo.row[sel], rng.integers(0, n_users, size=n_query - len(sel))]).astype(np.int64)
    items = np.concatenate([coo.col[sel], rng.integers(0, n_items, size=n_query - len(sel))]).astype(np.int64)
    # row ends of a few rows, and the neighbours of present items (off-by-one in the binary search)
    users = np.concatenate([users, [0, 0, 0, n_users // 2, n_users // 2, n_users - 1, 3, 3]])
    items = np.concatenate([items, [0, n_items - 1, 1, 0, 1, n_items - 1, 0, n_items - 1]])
    perm = rng.permutation(len(users))
    users, items = users[perm], items[perm]
    got = dpi.contains(users, items)
    assert got.dtype == bool and got.shape == users.shape
    assert np.array_equal(got, host.contains(users, items))
    for q in rng.integers(0, len(users), size=min(3000, len(users))):       # the reference's formulation, literally
        u, v = int(users[q]), int(items[q])
        assert bool(got[q]) == (v in m.indices[m.indptr[u]:m.indptr[u + 1]])
    assert got.sum() >= n_query // 2
    assert len(dpi.contains(users[:0], items[:0])) == 0


def _g6_dataset(w, meta):
    ds = S().SyntheticDataset.__new__(S().SyntheticDataset)
    ds.interaction_matrix = w['inter'].tocoo()
    ds.user_sampling_matrix = w['inter']
    ds.items_in_split = np.arange(I)
    ds.n_items = I
    ds.n_negative_samples = meta['n_neg']
    ds.negative_sampling_strategy = 'uniform_recbole'
    return ds


def test_g6_streams_through_the_gpu_collate(monkeypatch):
    """The reference's golden batches (G6: default ``uniform_recbole`` collate and the ``uniform`` collate, seed 42, shuffled epoch
    order) through ``NegativeSamplingDataLoader(device='cuda')`` with every membership round forced onto the GPU kernel: the
    native one-call host collate is switched off and ``HOST_BELOW`` is 0."""
    from oracle import sampling_ref
    monkeypatch.setenv('SBR_NATIVE_COLLATE', '0')
    z = load('g6_neg_sampling')
    meta = MANIFEST['g6_neg_sampling']
    ds = _g6_dataset(world(z), meta)
    for strategy, key in (('uniform_recbole', 'recbole'), ('uniform', 'uniform')):
        sampling_ref.reproducible(42)
        loader = S().NegativeSamplingDataLoader(ds, batch_size=meta['batch_size'], shuffle=True, strategy=strategy, device=DEV)
        assert isinstance(loader.positives, S().sampling.DevicePositiveIndex)
        loader.positives.HOST_BELOW = 0
        calls = []
        orig = loader.positives.contains
        loader.positives.contains = lambda u_, v_: (calls.append(len(u_)), orig(u_, v_))[1]
        for b, (u, i, l) in enumerate(loader):
            if b >= 3:
                break
            assert u.dtype == torch.int64 and i.dtype == torch.int64 and l.dtype == torch.float64
            assert (u.numpy() == z[f'{key}/u{b}']).all() and (i.numpy() == z[f'{key}/i{b}']).all()
            assert (l.numpy() == z[f'{key}/l{b}']).all()
        assert len(calls) >= 3


def test_large_batch_collate_on_gpu_equals_the_oracle_stream():
    """The collate path of the bench (B = 8192: 81,920 slots, beyond the native small-batch collate; first membership round on
    the GPU, the shrinking redraw rounds on the host copy) against the oracle's literal restatement of
    data/dataloader.py:154-198 on the same seeds: three consecutive batches, identical users, items and labels, and the
    global numpy generator left in the same state."""
    from oracle import sampling_ref
    ds = S().SyntheticDataset(20_000, 10_000, 1_000_000, seed=3, n_negative_samples=10)
    B = 8192
    sampling_ref.reproducible(7)
    loader = S().NegativeSamplingDataLoader(ds, batch_size=B, shuffle=True, device=DEV, max_batches=3)
    kinds = []
    orig = loader.positives.contains
    loader.positives.contains = lambda u_, v_: (kinds.append(len(u_)), orig(u_, v_))[1]
    got = [tuple(t.numpy().copy() for t in batch) for batch in loader]
    assert len(got) == 3
    after = np.random.randint(0, 1 << 30)
    assert max(kinds) == B * 10 and max(kinds) >= loader.positives.HOST_BELOW     # the big rounds ran on the GPU kernel
    # oracle: same seeds, same epoch order, literal python membership loop
    sampling_ref.reproducible(7)
    order = sampling_ref.loader_epoch_order(len(ds))
    coo = ds.interaction_matrix
    inter = ds.user_sampling_matrix
    positives = [inter.indices[inter.indptr[u]:inter.indptr[u + 1]] for u in range(ds.n_users)]
    for b in range(3):
        sel = order[b * B:(b + 1) * B]
        u, i, l = sampling_ref.recbole_collate(coo.row[sel], coo.col[sel], 10, ds.items_in_split, positives)
        assert np.array_equal(got[b][0], u) and np.array_equal(got[b][1], i) and np.array_equal(got[b][2], l)
    assert np.random.randint(0, 1 << 30) == after


# ---- b) the fused step against the golden groups ----------------------------------------------------------------------------------
_LOSS = {
    'bce': ('bce', 'mean', 'uniform_recbole'), 'bpr': ('bpr', 'mean', 'uniform_recbole'), 'bpr_sum': ('bpr', 'sum', 'uniform_recbole'),
    'ssm_uniform': ('sampled_softmax', 'mean', 'uniform'), 'ssm_recbole': ('sampled_softmax', 'sum', 'uniform_recbole'),
}


def _loss(name, n_neg=3):
    kind, agg, strat = _LOSS[name]
    return S().RecommenderSystemLossesEnum[kind].value(n_items=I, aggregator=agg, train_neg_strategy=strat, neg_train=n_neg)


def _draw_of(ent, names):
    """Recorded modality NAMES [*shape, k] -> the (positions, order) pair FusedTrainStep.step takes as a draw."""
    order = ent.train_modality_order
    lut = {m: i for i, m in enumerate(order)}
    pos = np.vectorize(lut.__getitem__, otypes=[np.int8])(np.asarray(names)).reshape(-1, names.shape[-1])
    return pos, order


def _draws(net, z, prefix, suffix=''):
    um = z[f'{prefix}user_mods{suffix}'] if f'{prefix}user_mods{suffix}' in z.files else None
    du = _draw_of(net.user_embedding_module, um) if um is not None else None
    return du, _draw_of(net.item_embedding_module, z[f'{prefix}item_mods{suffix}'])


@pytest.mark.parametrize('case', MANIFEST['g4_full_net']['cases'], ids=lambda c: c['name'])
def test_fused_step_against_g4_directly(case):
    """``FusedTrainStep.step`` on the reference's recorded batch and modality decisions: rec loss, reg loss, every parameter
    gradient (read from the flat gradient buffer right before the optimizer launch) and the BatchNorm running statistics
    against the golden values — three times: plain launches (first sighting of the signature), hipGraph capture + replay
    (second), replay (third). The optimizer launch is replaced by a recorder, so all three see the same parameters."""
    z = load('g4_full_net')
    n = case['name']
    net = product_net(z, case, f'{n}/sd0/')
    net.train()
    opt = S().FusedOptimizer(net, 'adamw', lr=1e-3, weight_decay=0.)
    fused = S().FusedTrainStep(net, _loss(case['loss']), opt, use_graph=True)
    seen = []
    def _record(*a, **k):                                      # stands in for the optimizer launch (which also resets the gradient)
        seen.append({k_: p.grad.detach().clone() for k_, p in net.named_parameters()})
        if k.get('zero_grad'):
            opt.fp.grad.zero_()
    opt.step_flat = _record
    u, i, labels = (torch.from_numpy(z[f'{n}/{k}']) for k in ('u', 'i', 'labels'))
    golden = sub(z, f'{n}/g/')
    sc = gscale(golden.values())
    for rep in range(3):
        total, rec, reg = fused.step(u, i, labels, _draws(net, z, f'{n}/'))
        close(rec.cpu(), z[f'{n}/rec_loss'].astype(np.float64), what=f'rec_loss (pass {rep})', rtol=1e-4, atol=1e-6)
        close(reg.cpu().reshape(-1), z[f'{n}/reg_loss'].astype(np.float64).reshape(-1), what=f'reg_loss (pass {rep})', rtol=1e-4, atol=1e-6)
        close(total.cpu(), float(z[f'{n}/rec_loss']) + float(np.asarray(z[f'{n}/reg_loss']).sum()), what='total', rtol=1e-4, atol=1e-6)
        assert len(seen) == rep + 1
        for k_, g in golden.items():
            close(seen[rep][k_].cpu(), g, what=f'grad {k_} (pass {rep})', rtol=2e-4, atol=1e-5, scale=sc, norm_rtol=1e-4)
        if rep == 0:
            for k_, v in sub(z, f'{n}/sd1/').items():
                close(net.state_dict()[k_].cpu(), v, what=f'sd1/{k_}', rtol=1e-4, atol=1e-5)
    assert fused.n_replays == 2 and float(opt.fp.grad.abs().max()) == 0.0          # the step re-zeroed the flat gradients
    fused.close()


@pytest.mark.parametrize('case', MANIFEST['g8_optim']['cases'], ids=lambda c: c['name'])
def test_fused_step_against_g8_trajectories(case):
    """Three real fused steps (forward, loss, hand-written backward, ONE fused optimizer launch) replaying the reference's
    batches and modality decisions: per-step losses and every parameter after step 3 against the reference's trajectory
    (AdamW / Adam / Adagrad with weight decay; rows of the tables that no batch touches included)."""
    z = load('g8_optim')
    n = case['name']
    net = product_net(z, case, f'{n}/sd0/')
    net.train()
    opt = S().FusedOptimizer(net, case['optimizer'], lr=case['lr'], weight_decay=case['wd'])
    loss_fn = S().RecBayesianPersonalizedRankingLoss(n_items=I, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
    fused = S().FusedTrainStep(net, loss_fn, opt)
    for s in range(3):
        u, i, labels = (torch.from_numpy(z[f'{n}/{k}{s}']) for k in ('u', 'i', 'labels'))
        total, rec, reg = fused.step(u, i, labels, _draws(net, z, f'{n}/', str(s)))
        close(rec.cpu(), z[f'{n}/loss{s}'].astype(np.float64), what=f'loss{s}', rtol=2e-4, atol=1e-5)
    final = sub(z, f'{n}/sd3/')
    skip = bn_shadowed_biases(final.keys())
    sd = net.state_dict()
    for k_, v in final.items():
        if k_ not in skip:
            close(sd[k_].cpu(), v, what=f'sd3/{k_}', rtol=2e-4, atol=2e-5, norm_rtol=1e-4)
    fused.close()


# ---- c) fused fp16 evaluation end to end ---------------------------------------------------------------------------------------------
class _Fp16Rounded(torch.nn.Module):
    """A model whose representations are already fp16 values (held in fp32): both scorers then see identical inputs, and the
    fp32 route computes the arithmetic the fused kernel promises (fp32 accumulation of exact fp16 products)."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def get_item_representations(self, i):
        return self.net.get_item_representations(i).half().float()

    def get_user_representations(self, u):
        return self.net.get_user_representations(u).half().float()

    def combine_user_item_representations(self, u, i):
        return self.net.combine_user_item_representations(u, i)

    def check_index_errors(self):
        self.net.check_index_errors()


def _eval(alg, view, scorer, top_k=(1, 10, 20), **kw):
    ev = S().FullEvaluator(config=S().evaluation._Cfg(top_k=top_k, calculate_std=False), dataset=view)
    loader = type('L', (), {'dataset': view, 'batch_size': 64})()
    return S().evaluate_recommender_algorithm(alg, loader, ev, DEV, return_raw=True, scorer=scorer, **kw)


def _assert_same_metrics(a, b, what, tie_users=0):
    """Per-user metric arrays and their means must be equal. ``tie_users``: how many users may differ because two of their
    scores tie to the last bit — the fp32 GEMM adds the (exact) fp16 products in k order, the MFMA in its own order, so among
    ~10^5 neighbouring pairs of a large world a handful of 1-ulp ties can come out in the other order."""
    (ma, ra), (mb, rb) = a, b
    assert list(ma) == list(mb), what
    for k in ra:
        n_diff = int((ra[k] != rb[k]).sum())
        assert n_diff <= tie_users, f'{what}: per-user {k} differs for {n_diff} users'
    for k in ma:
        if tie_users == 0:
            assert ma[k] == mb[k], f'{what}: {k} {ma[k]} vs {mb[k]}'
        else:
            assert abs(ma[k] - mb[k]) <= 2e-6 * tie_users, f'{what}: {k} {ma[k]} vs {mb[k]}'


def _g9_view(z):
    w = world(z)
    return SimpleNamespace(n_users=U, n_items=I, items_in_split=np.arange(I), users_in_split=np.arange(U), n_items_in_split=I,
                           n_users_in_split=U, user_sampling_matrix=sp.csr_matrix(z['labels']), exclude_data=w['inter'].astype(bool))


def test_g9_through_evaluate_recommender_algorithm_both_scorers():
    """The golden evaluation world through the public function. The golden model has D = 8, outside the fused kernel's
    dimensions: ``scorer='fp16_fused'`` must take its fp32 fall-back and both requests must reproduce the reference's
    per-user NDCG / recall / precision (eval/metrics.py definitions) exactly as ``test_g9_eval_fp32_path`` pins them."""
    z = load('g9_eval')
    net = product_net(z, MANIFEST['g9_eval'], 'sd/')
    view = _g9_view(z)
    res = {s: _eval(net, view, s) for s in ('fp32', 'fp16_fused')}
    _assert_same_metrics(res['fp32'], res['fp16_fused'], 'odd-D fall-back')
    metrics, raw = res['fp16_fused']
    for k in (1, 10, 20):
        for name in ('ndcg', 'recall', 'precision'):
            close(raw[f'{name}@{k}'], z[f'{name}@{k}'], what=f'{name}@{k}', rtol=1e-5, atol=1e-6)
            assert abs(metrics[f'{name}@{k}'] - float(z[f'{name}@{k}'].mean())) < 1e-6


def _world_net(n_users, n_items, nnz, D, seed=5, train_steps=0):
    ds = S().SyntheticDataset(n_users, n_items, nnz, item_dense={'text': 48}, item_tags={'genres': (12, 3)}, seed=seed,
                              n_negative_samples=5, holdout_per_user=2)
    cfg = {'shared_common_dim': D, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'genres'}, {'feature_name': 'item_embedding'}],
                    'single_branch_hidden_layers': [D], 'preference_hidden_layers': [], 'common_modality_dim': D}}
    torch.manual_seed(seed)
    np.random.seed(seed)
    net = S().SingleBranchNet(S().SingleBranchNetConfig.from_dict(cfg), ds).to(DEV)
    if train_steps:
        net.train()
        loss = S().RecBayesianPersonalizedRankingLoss(n_items=n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=5)
        fused = S().FusedTrainStep(net, loss, S().FusedOptimizer(net, 'adamw', lr=3e-3, weight_decay=1e-6))
        loader = S().NegativeSamplingDataLoader(ds, batch_size=2048, shuffle=True, device=DEV, max_batches=train_steps)
        for b in loader:
            fused.step(*b)
        fused.close()
    net.eval()
    return ds, net


@pytest.mark.parametrize('D', [64, 128, 256])
def test_fused_evaluation_equals_fp32_evaluation_on_the_golden_world(D):
    """The G9 world (50 users x 40 items, its exclusion and label matrices) with a D-wide model: the route cast_f16 -> fused
    score + mask + top-k -> rank metrics gives, user by user, the same NDCG / recall / precision / hit rate / F-score and the
    same coverage as GEMM -> mask -> exact top-k on the same fp16-rounded representations. 40 items: catalogue shorter than
    a tile and cut-off 20 = half the catalogue."""
    z = load('g9_eval')
    view = _g9_view(z)
    ds = SimpleNamespace(n_users=U, n_items=I, user_features={}, item_features={
        'text': S().HostFeature('text', 'dense', world(z)['text'])}, user_sampling_matrix_train=world(z)['inter'],
        item_sampling_matrix_train=world(z)['inter_t'], is_cold_start_user=False, is_cold_start_item=False)
    cfg = {'shared_common_dim': D, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'interactions'}], 'single_branch_hidden_layers': [D],
                    'preference_hidden_layers': [], 'common_modality_dim': D}}
    torch.manual_seed(D)
    net = S().SingleBranchNet(S().SingleBranchNetConfig.from_dict(cfg), ds).to(DEV).eval()
    alg = _Fp16Rounded(net)
    _assert_same_metrics(_eval(alg, view, 'fp32'), _eval(alg, view, 'fp16_fused'), f'golden world, D = {D}')


def test_fused_evaluation_on_a_20k_user_world():
    """20k users x 6k items, D = 128, a briefly trained model (so that scores are structured, not noise):
      * on fp16-rounded representations the fused and the fp32 evaluation agree user by user (identical NDCG@10 and all else);
      * chunked launches (user_chunk = 7000: 3 launches, the last ragged) equal the single launch;
      * on the unrounded model the fused scorer's NDCG@10 is the fp32 scorer's within 1 % + 1e-4 (fp16 rounding of the
        representations moves a few near-ties; 20k users: one user's hit is 5e-5 of a mean) — the "matched NDCG@10" of the bench's scores/s figure;
      * cut-offs beyond 32 take the fp32 route and return what the fp32 scorer returns."""
    ds, net = _world_net(20_000, 6_000, 400_000, 128, train_steps=40)
    view = ds.eval_view()
    alg = _Fp16Rounded(net)
    fp32 = _eval(alg, view, 'fp32')
    fused = _eval(alg, view, 'fp16_fused')
    _assert_same_metrics(fp32, fused, 'rounded representations', tie_users=3)
    assert fp32[0]['ndcg@10'] > 0
    _assert_same_metrics(fused, _eval(alg, view, 'fp16_fused', user_chunk=7000), 'chunked launches')
    a, b = _eval(net, view, 'fp32')[0], _eval(net, view, 'fp16_fused')[0]
    for k in ('ndcg@10', 'recall@10', 'precision@10', 'ndcg@20'):
        assert abs(a[k] - b[k]) <= 1e-2 * a[k] + 1e-4, (k, a[k], b[k])
    wide = (1, 10, 50)
    _assert_same_metrics(_eval(alg, view, 'fp32', top_k=wide), _eval(alg, view, 'fp16_fused', top_k=wide), 'k > 32 fall-back')


# ---- d) BASELINE configs at their own shapes ------------------------------------------------------------------------------------------
def test_c3_step_at_full_shapes_against_the_cpu_oracle():
    """BASELINE configs[2] (Onion18 shape): 13,610 items with a 1024-d dense modality, an 853-tag bag and the CSR interactions
    modality (5,192 columns), C = 512, five hidden layers [512, 512, 512, 256, 256], D = 128, pairwise InfoNCE (tau 0.1),
    BPR, batch 256 x 11 -> 5,632 rows through the shared network. One fused step against the CPU oracle on the same parameters,
    batch and modality draw: rec loss, reg loss and every parameter gradient."""
    from oracle import losses_ref, model_ref
    ds = S().SyntheticDataset(5192, 13610, 326_000, item_dense={'audio': 1024}, item_tags={'genres': (853, 5)}, seed=0,
                              n_negative_samples=10)
    cfg = {'shared_common_dim': 128, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'interactions'}, {'feature_name': 'genres'}, {'feature_name': 'audio'}],
                    'single_branch_hidden_layers': [512, 512, 512, 256, 256], 'preference_hidden_layers': [],
                    'common_modality_dim': 512, 'embedding_regularization_type': 'pairwise_single',
                    'regularization_temperature': 0.1, 'regularization_weight': 1e-4}}
    torch.manual_seed(42)
    np.random.seed(42)
    net = S().SingleBranchNet(S().SingleBranchNetConfig.from_dict(cfg), ds).to(DEV).train()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and 'running' not in k:
            v.requires_grad_(True)
    opt = S().FusedOptimizer(net, 'adamw', lr=1e-3, weight_decay=0.)
    lossf = S().RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
    fused = S().FusedTrainStep(net, lossf, opt)
    seen = []
    def _record(*a, **k):                                      # stands in for the optimizer launch (which also resets the gradient)
        seen.append({k_: p.grad.detach().clone() for k_, p in net.named_parameters()})
        if k.get('zero_grad'):
            opt.fp.grad.zero_()
    opt.step_flat = _record
    loader = S().NegativeSamplingDataLoader(ds, batch_size=256, shuffle=True)
    u, i, labels = next(iter(loader))
    draws = fused.draw(u.shape, i.shape)
    import importlib
    _lib = importlib.import_module(S().ops.__name__.rsplit('.', 1)[0] + '._lib')
    _lib.CALL_LOG = []
    total, rec, reg = fused.step(u, i, labels, draws)
    names, _lib.CALL_LOG = [n_ for n_, _ in _lib.CALL_LOG], None
    # 5,632 rows: the 256 -> 128 output layer on the bf16-split projector kernel, the weight gradients on the bf16-split dW kernel; the
    # 512- and 256-wide forward / input-gradient products stay on the fp32 ring kernel at this batch (22 x 2 tiles of 256 x 256 do not
    # fill the chip: ops._wide_ok) and move to the wide bf16-split kernel from ~22k rows on
    assert 'sbr_gemm_split_proj_f32' in names and names.count('sbr_gemm_tn_f32_slabs') >= 6, names
    pos, order = draws[1]
    mods = np.array(order)[pos].reshape(tuple(i.shape) + (2,))
    ut = {'user_embedding': model_ref.RefTable('categorical', np.arange(ds.n_users), n_categories=ds.n_users)}
    it = {k: model_ref.table_from_feature(f) for k, f in ds.item_features.items()}
    ref = model_ref.RefSingleBranchNet(sd, cfg, ut, it, orders={'item_train': net.item_embedding_module.train_modality_order,
                                                                'item_eval': net.item_embedding_module.eval_modality_order})
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    logits = ref.forward(u, i, True, None, mods)
    rl = losses_ref.bpr_loss(logits, labels)
    rr = ref.get_and_reset_other_loss()['reg_loss']
    (rl + rr.sum()).backward()
    close(rec.cpu(), rl.detach().double(), what='rec loss', rtol=1e-4, atol=1e-6)
    close(reg.cpu().reshape(-1), rr.detach().double().reshape(-1), what='reg loss', rtol=1e-4, atol=1e-7)
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    sc = gscale(grads.values())
    # six GEMM layers of width 512 with five BatchNorms between the loss and the first projector: the fp32 rounding of two
    # summation orders (torch-CPU's and the MFMA k-order) separates single elements by a few 1e-4 of the gradient scale
    # (measured: 2.5e-4 on one bias element); losses stay within the north star's 1e-4
    for k_, g in grads.items():
        close(seen[0][k_].cpu(), g, what=f'grad {k_}', rtol=5e-4, atol=1e-6, scale=sc, norm_rtol=5e-4)
    fused.close()


def test_c2_step_at_the_bench_batch_against_the_cpu_oracle():
    """BASELINE configs[1] exactly as ``bench.py`` times it: the bench's model (bench.build), one loader batch of 8,192 interactions
    x 11 slots with its recorded modality draw, ``FusedTrainStep.step`` five times — three passes of plain launches (the arena is sized, grown,
    re-sighted), hipGraph capture + replay, replay — with the optimizer launch replaced by a recorder, against the CPU oracle (oracle/model_ref.py restating
    train/trainer.py:204-223 -> sgd_alg.py:2116-2125, rec_losses.py:88-113) on the same parameters, batch and draw: the
    sampled-softmax loss (1e-4 relative) and EVERY gradient incl. both embedding tables (norm-wise 1e-4). The call log of the
    plain-launch pass must show the launch mix the bench line is timed on: bf16-split projector, bf16-split K = N = 128 products
    (forward with the BatchNorm statistics epilogue, backward), bf16-split dW products, the fused scorer + loss + statistics kernel,
    the one-launch lookup."""
    import importlib
    from oracle import losses_ref, model_ref
    bench = importlib.import_module('bench')
    _lib = importlib.import_module(S().ops.__name__.rsplit('.', 1)[0] + '._lib')
    ds, net = bench.build(S(), dict(bench.C2), DEV)
    net.train()
    cfg = bench.model_config(bench.C2['emb_dim'])
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and 'running' not in k:
            v.requires_grad_(True)
    opt = S().FusedOptimizer(net, 'adamw', lr=1e-3, weight_decay=0.)
    lossf = S().RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole',
                                      neg_train=ds.n_negative_samples)
    fused = S().FusedTrainStep(net, lossf, opt, use_graph=True)
    seen = []
    def _record(*a, **k):                                      # stands in for the optimizer launch (which also resets the gradient)
        seen.append({k_: p.grad.detach().clone() for k_, p in net.named_parameters()})
        if k.get('zero_grad'):
            opt.fp.grad.zero_()
        return False
    opt.step_flat = _record
    np.random.seed(42)
    loader = S().NegativeSamplingDataLoader(ds, batch_size=8192, shuffle=True)
    u, i, labels = next(iter(loader))
    assert tuple(i.shape) == (8192, 11)
    draws = fused.draw(u.shape, i.shape)
    recs = []
    n_rep = 5            # plain launches: sizing the arena, growing it, first sighting in the grown arena; then capture + replay, replay
    for rep in range(n_rep):
        if rep == 0:
            _lib.CALL_LOG = []
        total, rec, reg = fused.step(u, i, labels, draws)
        if rep == 0:
            log, _lib.CALL_LOG = _lib.CALL_LOG, None
        recs.append(rec.cpu())
    assert fused.n_replays == 2 and len(seen) == n_rep
    # ---- the launch mix of the timed path
    names = [n_ for n_, _ in log]
    lib = _lib.lib()
    assert names.count('sbr_gemm_split_proj_f32') == 1, names                       # projector forward 45k x 128 x 768, gathered
    assert names.count('sbr_gemm_split_bnstats_f32') == 1                           # last Linear + BatchNorm statistics + finalisation
    assert names.count('sbr_gemm_split_f32') >= 1                                   # dX of the shared layer (K = N = 128)
    assert 'sbr_gemm_wres_f32' not in names and 'sbr_gemm_f32' not in names         # nothing on the fp32 pipe
    tn = [a for n_, a in log if n_ == 'sbr_gemm_tn_f32_slabs']
    assert len(tn) == 3 and all(lib.sbr_gemm_tn_split_supported(int(a[6]), int(a[7]), int(a[8])) for a in tn), tn
    assert names.count('sbr_bn_score_loss_fwd_bwd') == 1 and names.count('sbr_bn_score_bwd_apply') == 1
    assert names.count('sbr_lookup_rows') == 1 and 'sbr_rec_loss_fwd_bwd' not in names
    # ---- the oracle on the same inputs
    pos, order = draws[1]
    mods = np.array(order)[pos].reshape(tuple(i.shape) + (1,))
    ut = {'user_embedding': model_ref.RefTable('categorical', np.arange(ds.n_users), n_categories=ds.n_users)}
    it = {'text': model_ref.table_from_feature(ds.item_features['text']),
          'item_embedding': model_ref.RefTable('categorical', np.arange(ds.n_items), n_categories=ds.n_items)}
    ref = model_ref.RefSingleBranchNet(sd, cfg, ut, it, orders={'item_train': net.item_embedding_module.train_modality_order,
                                                                'item_eval': net.item_embedding_module.eval_modality_order})
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    logits = ref.forward(u, i, True, None, mods)
    rl = losses_ref.RefRecLoss('sampled_softmax', n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole',
                               neg_train=ds.n_negative_samples).compute_loss(logits, labels)
    rl.backward()
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    assert {'user_embedding_module.embedding_layer.weight'} <= set(grads) and len(grads) == len(seen[0])
    sc = gscale(grads.values())
    for rep in range(n_rep):
        close(recs[rep], rl.detach().double(), what=f'rec loss (pass {rep})', rtol=1e-4, atol=1e-7)
        for k_, g in grads.items():
            close(seen[rep][k_].cpu(), g, what=f'grad {k_} (pass {rep})', rtol=2e-4, atol=1e-7, scale=sc, norm_rtol=1e-4)
    fused.close()


@pytest.mark.parametrize('deferred', ['0', '1'])
def test_c4_step_on_one_gpu_at_full_table_shapes(deferred, monkeypatch):
    """(deferred = '1': the optimizer launch updates the 1M-row user table row by row — engine.DeferredTable, the default — and the
    table is flushed before it is read; '0': one dense launch over all 257 M parameters.)
    BASELINE configs[3] on ONE GPU (the 8-GPU job runs this per rank): 1M users x 200k items, text 768 + image 2048, C = D =
    256, user = embedding lookup (257 M parameters, 1 GB user table), sampled softmax, batch 256.
      * the fused step (plain launches, then capture + replay) against the module / autograd path over the same kernels, which
        the golden groups pin: loss and every gradient — all 257 M elements of the flat gradient buffer;
      * one real step: the single dense AdamW launch over the 257 M parameters against the update rule (oracle/train_ref.py)
        applied to that gradient — touched rows, a stride sample of untouched rows (pure weight decay) and two checksums of the
        whole user table."""
    from oracle import train_ref
    monkeypatch.setenv('SBR_DEFERRED_ADAM', deferred)
    ds = S().SyntheticDataset(1_000_000, 200_000, 4_000_000, item_dense={'text': 768, 'image': 2048}, seed=0, n_negative_samples=10)
    cfg = {'shared_common_dim': 256, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'image'}], 'single_branch_hidden_layers': [256],
                    'preference_hidden_layers': [], 'common_modality_dim': 256}}
    nets = []
    for _ in range(2):
        torch.manual_seed(42)
        np.random.seed(42)
        nets.append(S().SingleBranchNet(S().SingleBranchNetConfig.from_dict(cfg), ds).to(DEV).train())
    assert sum(p.numel() for p in nets[0].parameters()) > 256_000_000
    lr, wd = 1e-2, 1e-2
    opts = [S().FusedOptimizer(n_, 'adamw', lr=lr, weight_decay=wd) for n_ in nets]
    lossf = S().RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
    fused = S().FusedTrainStep(nets[1], lossf, opts[1])
    real_step = opts[1].step_flat
    seen = []
    def _record(*a, **k):                                      # stands in for the optimizer launch (which also resets the gradient)
        seen.append(opts[1].fp.grad.clone())
        if k.get('zero_grad'):
            opts[1].fp.grad.zero_()
    opts[1].step_flat = _record
    rng = np.random.default_rng(8)
    u = torch.from_numpy(rng.integers(0, ds.n_users, size=256))
    i = torch.from_numpy(rng.integers(0, ds.n_items, size=(256, 11)))
    labels = torch.zeros(256, 11, dtype=torch.float64)
    labels[:, 0] = 1
    draws = fused.draw(u.shape, i.shape)
    pos, order = draws[1]
    mods = np.array(order)[pos].reshape(tuple(i.shape) + (1,))
    logits = nets[0](u.to(DEV), i.to(DEV), item_modalities=mods)
    loss = lossf.compute_loss(logits, labels.to(DEV))
    loss.backward()
    opts[0]._sync_grads()
    g_ref = opts[0].fp.grad
    scale = float(g_ref.abs().max())
    assert scale > 0
    for rep in range(4):           # plain launches: sizing the arena, growing it, first sighting in the grown arena; then capture + replay
        total, rec, reg = fused.step(u, i, labels, draws)
        assert torch.isfinite(total).all()
        close(rec.cpu(), loss.detach().cpu().double(), what=f'rec loss (pass {rep})', rtol=2e-5, atol=1e-7)
        err = float((seen[rep] - g_ref).abs().max())
        assert err <= 1e-4 * scale, f'flat gradient (pass {rep}): max abs err {err:.3e} at scale {scale:.3e}'
    assert fused.n_replays == 1
    # one real optimizer launch over all 257 M parameters
    opts[1].step_flat = real_step
    p0 = {k_: v.detach().clone() for k_, v in nets[1].named_parameters()}
    fused.step(u, i, labels, draws)
    assert (fused.deferred is not None) == (deferred == '1')
    fused.flush()                                              # a row-wise updated table is brought up to date before it is read
    fp = opts[1].fp
    name_of = {id(p): k_ for k_, p in nets[1].named_parameters()}
    rows = torch.cat([u, torch.arange(0, 1_000_000, 9973)]).to(DEV)
    for p, o, n_ in zip(fp.params, fp.offsets, fp.sizes):
        k_ = name_of[id(p)]
        g = torch.as_strided(seen[3], p.shape, p.stride(), o)
        want, _, _ = train_ref.adamw_update(p0[k_].double(), g.double(), torch.zeros_like(g).double(), torch.zeros_like(g).double(),
                                            1, lr, wd)
        if k_ == 'user_embedding_module.embedding_layer.weight':
            close(p.detach()[rows].cpu(), want[rows].cpu(), what=f'{k_} (touched + sampled rows)', rtol=1e-5, atol=1e-7)
            assert abs(float(p.detach().double().sum() - want.sum())) <= 1e-6 * float(want.abs().sum())
            assert abs(float(p.detach().double().pow(2).sum() - want.pow(2).sum())) <= 1e-6 * float(want.pow(2).sum())
        else:
            # elements whose gradient is rounding noise around zero take +-lr steps of either sign under Adam (golden_util.
            # bn_shadowed_biases): the replay's gradient bits equal pass 1's only up to summation order, so compare where
            # the gradient is clearly non-zero and bound the rest by the step size
            solid = g.abs() > 1e-6 * scale
            close(torch.where(solid, p.detach().double(), want).cpu(), want.cpu(), what=k_, rtol=1e-4, atol=1e-6)
            assert float((p.detach().double() - want).abs().max()) <= 2.5 * lr
    fused.close()


def test_c5_shard_shape_fused_scorer_against_fp32_gemm_topk():
    """BASELINE configs[4], one of its eight item shards at its own size: 100k users x 25k items x 256 fp16, 50 excluded items
    per user, top-20, shard offset 75,000 (the fourth... last-but-four shard's global indices). The fused kernel's lists for a
    4,096-user sample (taken from the full 100k-user launch, so the workgroup geometry is the real one) against the fp32 MFMA
    GEMM -> CSR mask -> exact top-k route on the same fp16 values; every list is sorted, global, and free of excluded items."""
    ops = S().ops
    g = torch.Generator(device=DEV).manual_seed(7)
    n_u, n_i, D, k, off = 100_000, 25_000, 256, 20, 75_000
    u16 = ops.cast_f16(torch.randn(n_u, D, device=DEV, generator=g) / 16)
    i16 = ops.cast_f16(torch.randn(n_i, D, device=DEV, generator=g) / 16)
    rng = np.random.default_rng(3)
    # exclusions in GLOBAL item ids over the whole 200k catalogue: ~50 per user, ~6 of them inside this shard
    cols = np.sort(rng.integers(0, 200_000, size=(n_u, 50)), axis=1)
    indptr = np.arange(0, 50 * n_u + 1, 50, dtype=np.int64)
    m = sp.csr_matrix((np.ones(cols.size, dtype=np.int8), cols.reshape(-1), indptr), shape=(n_u, 200_000))
    m.sum_duplicates()
    m.sort_indices()
    eptr, eidx = torch.from_numpy(m.indptr.astype(np.int64)).to(DEV), torch.from_numpy(m.indices.astype(np.int32)).to(DEV)
    users = torch.arange(n_u, device=DEV)
    val, idx = ops.score_topk_f16(u16, i16, k, users, eptr, eidx, item_offset=off)
    assert (idx >= off).all() and (idx < off + n_i).all()
    assert (val[:, :-1] >= val[:, 1:]).all()
    sample = torch.from_numpy(np.sort(rng.choice(n_u, size=4096, replace=False))).to(DEV)
    sc = ops.linear_nt(u16[sample].float(), i16.float())
    shard = sp.csr_matrix(m[:, off:off + n_i])
    shard.sort_indices()
    ops.mask_scores_(sc, sample, torch.from_numpy(shard.indptr.astype(np.int64)).to(DEV), torch.from_numpy(shard.indices.astype(np.int32)).to(DEV))
    tv, ti = ops.topk_rows(sc, k)
    close(val[sample].cpu(), tv.cpu(), what='top-k values', rtol=1e-5, atol=1e-6)
    same = (idx[sample] - off == ti)
    # equal scores may swap places between an MFMA-fp16 and an fp32-GEMM summation order only when they tie to the last bit
    assert same.float().mean().item() > 0.9995
    got = torch.gather(sc, 1, (idx[sample] - off).long())
    close(got.cpu(), tv.cpu(), what='scores of the selected items', rtol=1e-5, atol=1e-6)
    assert not torch.isinf(got).any()


# ---- e) FullEvaluator group metrics (eval/eval.py:74-92, 106-119) -----------------------------------------------------------------------
def test_group_metrics_on_the_golden_world():
    """Group-wise metrics per categorical user feature: the G9 world carries 'gender' and 'age'. The evaluator's per-group arrays
    must be the oracle's restriction (oracle/eval_ref.group_metrics) of the REFERENCE's golden per-user arrays to the users of
    each label; means, stds and the natsorted key order follow; an evaluator name prefixes every key; nine cut-offs take two
    launches of the metric kernel."""
    from oracle import eval_ref
    z = load('g9_eval')
    w = world(z)
    net = product_net(z, MANIFEST['g9_eval'], 'sd/')
    view = _g9_view(z)
    view.user_features = {
        'gender': S().HostFeature('gender', 'categorical', w['gender'], n_categories=w['gender_ncat'],
                                  unique_values=['F', 'M', 'X'][:w['gender_ncat']] if w['gender_ncat'] <= 3 else None),
        'age': S().HostFeature('age', 'categorical', w['age'], n_categories=w['age_ncat']),
        'user_embedding': S().HostFeature('user_embedding', 'categorical', np.arange(U), n_categories=U)}
    cfg = S().evaluation._Cfg(top_k=(1, 10, 20), metrics=['ndcg', 'recall', 'precision'], calculate_std=True,
                              calculate_group_metrics=True)
    ev = S().FullEvaluator(config=cfg, evaluator_name='val', dataset=view)
    assert ev._user_features == ['gender', 'age']
    loader = type('L', (), {'dataset': view, 'batch_size': 16})()
    metrics, raw = S().evaluate_recommender_algorithm(net, loader, ev, DEV, return_raw=True, user_chunk=16)   # 4 user batches
    per_user = {k: z[k] for k in z.files if '@' in k}
    want = {}
    for name in ('gender', 'age'):
        labels = view.user_features[name].get_labels(w[name])
        want.update(eval_ref.group_metrics(per_user, name, labels))
    assert len(want) == 9 * (len(set(w['gender'].tolist())) + len(set(w['age'].tolist())))
    for k, v in want.items():
        close(raw[f'val/{k}'], v, what=k, rtol=1e-5, atol=1e-6)
        assert abs(metrics[f'val/{k}'] - float(v.mean())) < 1e-6 and abs(metrics[f'val/{k}_std'] - float(v.std())) < 1e-6
    assert 'val/gender_f/ndcg@10' in metrics or w['gender_ncat'] > 3
    assert list(metrics) == eval_ref.natural_sorted(list(metrics))
    assert set(raw) == {f'val/{k}' for k in want} | {f'val/{k}' for k in per_user}
    # more cut-offs than one launch of the metric kernel takes
    many = S().evaluation._Cfg(top_k=(1, 2, 3, 4, 5, 6, 10, 15, 20), metrics=['ndcg'], calculate_std=False)
    res = S().evaluate_recommender_algorithm(net, loader, S().FullEvaluator(config=many, dataset=view), DEV)
    assert [k for k in res] == [f'ndcg@{k}' for k in (1, 2, 3, 4, 5, 6, 10, 15, 20)]
    for k in (1, 10, 20):
        assert abs(res[f'ndcg@{k}'] - float(z[f'ndcg@{k}'].mean())) < 1e-6


# ---- f) the native batch producer (csrc/producer.hip) against the Python formulation of the same host work -----------------------------
@pytest.mark.parametrize('world,user_entity,reg', [(1, False, 'no_regularization'), (1, True, 'pairwise_single'),
                                                   (2, False, 'central_modality')])
def test_native_producer_batches_equal_the_python_pipeline(world, user_entity, reg, monkeypatch):
    """Every batch the C++ producer thread uploads — users, items with their negatives (MT19937 stream, membership rounds on the
    GPU kernel and on the host copy), the modality draws of both sides (PCG64 replica), the dropout seed, the padded per-modality
    counts of the launch plan — equals what the Python pipeline (sampling.recbole_negative_collate + FusedTrainStep.prepare, pinned
    by the golden streams G6 / G7) produces from the same generator states; including the ragged last batch of the epoch, a
    second epoch, and the states of numpy's global generator, of the entities' generators and of the seed counter afterwards.
    world = 2: rank-local sampling (rank 1's slices)."""
    ds = S().SyntheticDataset(700, 300, 5000, item_dense={'text': 24}, item_tags={'genres': (9, 3)}, user_categorical={'gender': 3},
                              seed=4, n_negative_samples=5)
    item = {'features': [{'feature_name': 'text'}, {'feature_name': 'genres'}, {'feature_name': 'item_embedding'}],
            'single_branch_hidden_layers': [16], 'preference_hidden_layers': [], 'common_modality_dim': 16,
            'embedding_regularization_type': reg, 'regularization_weight': 0.1}
    if reg == 'central_modality':
        item['central_modality'] = 'genres'
    user = ({'features': [{'feature_name': 'interactions'}, {'feature_name': 'gender'}], 'single_branch_hidden_layers': [],
             'preference_hidden_layers': [], 'common_modality_dim': 16, 'embedding_regularization_type': 'pairwise_single',
             'regularization_weight': 0.02} if user_entity else {'feature_name': 'user_embedding', 'embedding_dim': -1})
    cfg = {'shared_common_dim': 16, 'user': user, 'item': item}
    B = 512
    runs = {}
    for native in ('0', '1'):
        monkeypatch.setenv('SBR_NATIVE_PRODUCER', native)
        torch.manual_seed(21)
        np.random.seed(21)
        net = S().SingleBranchNet(S().SingleBranchNetConfig.from_dict(cfg), ds).to(DEV).train()
        lossf = S().RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=5)
        fused = S().FusedTrainStep(net, lossf, S().FusedOptimizer(net, 'adamw', lr=1e-3, weight_decay=0.))
        loader = S().NegativeSamplingDataLoader(ds, batch_size=B, shuffle=True, device=DEV, prefetch=3, prepare_fn=fused.prepare,
                                                rank=world - 1, world=world, dp_sampling='local')
        loader.positives.HOST_BELOW = 600            # first rounds (2,560 pairs) on the GPU kernel, redraw rounds on the host
        got = []
        for _ in range(2):                               # two epochs: the states handed back feed the next epoch
            for batch in loader:
                pb = batch[3]
                if pb.native is not None:
                    pb.native[0].wait(pb.native[1])
                elif pb.event is not None:
                    torch.cuda.current_stream().wait_event(pb.event)
                torch.cuda.synchronize()
                got.append(dict(u=pb.u.cpu().numpy().copy(), i=pb.i.cpu().numpy().copy(), si=pb.si.cpu().numpy().copy(),
                                su=None if pb.su is None else pb.su.cpu().numpy().copy(), seed=int(pb.seed.cpu()[0]),
                                pu=None if pb.pu is None else pb.pu[1:], pi=pb.pi[1:], shapes=(pb.u_shape, pb.i_shape)))
                if pb.native is not None:
                    pb.native[0].release(pb.native[1])
        assert (loader._native is not None) == (native == '1')
        loader.close()
        ents = ([net.user_embedding_module] if user_entity else []) + [net.item_embedding_module]
        runs[native] = (got, np.random.randint(0, 1 << 30), [int(e._rng.integers(0, 1 << 30)) for e in ents], fused._n_prepared)
    py, nat = runs['0'], runs['1']
    n_per_epoch = len(ds) // (B * world) if world > 1 else -(-len(ds) // B)
    assert len(py[0]) == len(nat[0]) == 2 * n_per_epoch
    if world == 1:
        assert py[0][n_per_epoch - 1]['shapes'][0][0] == len(ds) % B          # the ragged last batch of the epoch
    for b, (a, c) in enumerate(zip(py[0], nat[0])):
        assert a['shapes'] == c['shapes'] and a['pi'] == c['pi'] and a['pu'] == c['pu'], (b, a['pi'], c['pi'])
        assert a['seed'] == c['seed']
        for k_ in ('u', 'i', 'si', 'su'):
            assert (a[k_] is None and c[k_] is None) or np.array_equal(a[k_], c[k_]), (b, k_)
    assert py[1:] == nat[1:], 'generator states / seed counter after two epochs'


# ---- g) N-rank numerical parity ("N-GPU vs N x local batch, local BN emulated on CPU") ---------------------------
@pytest.mark.parametrize('case_name,sparse', [('adamw_lookup', '1'), ('adamw_lookup', '0'), ('adamw_entity', '1')])
def test_two_rank_data_parallel_parity_with_the_oracle(case_name, sparse, tmp_path):
    """Two ranks (one process each, sharing this box's GPU, gloo) run three fused data-parallel steps from the golden G8 state:
    rank r takes rows [r::2] of the reference's recorded batches and modality decisions, BatchNorm statistics stay rank-local,
    gradients are averaged over the ranks (dense all-reduce, or the all-gather of (row, gradient) pairs for the lookup user table)
    and every replica takes the same dense optimizer step. The CPU oracle emulates exactly that — each shard through
    RefSingleBranchNet with its own batch statistics, shard-mean losses, gradients averaged, torch.optim step — and every
    parameter of BOTH replicas must end where the emulation ends (tolerance of the G8 trajectories); the replicas agree bit for
    bit; rank 0's running statistics are the emulation's rank-0 statistics."""
    import subprocess
    import sys
    from oracle import losses_ref, model_ref, train_ref
    from golden_util import ref_tables, side_cfg
    here = os.path.dirname(os.path.abspath(__file__))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR')}
    env['SBR_SPARSE_EXCHANGE'] = sparse
    out = str(tmp_path / 'final')
    procs = [subprocess.Popen([sys.executable, os.path.join(here, 'dp_parity_worker.py'), case_name, str(r), '2', str(tmp_path / 'rdzv'), out],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(l[-1500:] for l in logs)
    got = [np.load(out + f'.rank{r}.npz') for r in range(2)]
    is_lookup = case_name.endswith('lookup')
    assert str(got[0]['exchange']) == ('sparse' if (sparse == '1' and is_lookup) else 'dense')
    # ---- oracle emulation
    z = load('g8_optim')
    case = [c for c in MANIFEST['g8_optim']['cases'] if c['name'] == case_name][0]
    cfg = {'shared_common_dim': case['shared_common_dim'], 'user': side_cfg(case['user']), 'item': side_cfg(case['item'])}
    orders = {k: case[k2] for k, k2 in [('user_train', 'user_train_order'), ('user_eval', 'user_eval_order'),
                                        ('item_train', 'item_train_order'), ('item_eval', 'item_eval_order')] if k2 in case}
    ut, it = ref_tables(world(z))
    sd = state_dict(z, f'{case_name}/sd0/', requires_grad=True)
    params = {k: v for k, v in sd.items() if v.requires_grad}
    opt = train_ref.make_optimizer(case['optimizer'], list(params.values()), case['lr'], case['wd'])
    stats = [{k: v.clone() for k, v in sd.items() if not v.requires_grad} for _ in range(2)]       # rank-local BatchNorm buffers
    for s in range(3):
        grads = {k: torch.zeros_like(v) for k, v in params.items()}
        for r in range(2):
            sd_r = dict(params)
            sd_r.update(stats[r])
            ref = model_ref.RefSingleBranchNet(sd_r, cfg, ut, it, orders=orders)
            u, i, labels = (torch.from_numpy(z[f'{case_name}/{k}{s}'])[r::2] for k in ('u', 'i', 'labels'))
            key = f'{case_name}/user_mods{s}'
            um = z[key][r::2] if key in z.files else None
            logits = ref.forward(u, i, True, um, z[f'{case_name}/item_mods{s}'][r::2])
            loss = losses_ref.bpr_loss(logits, labels) + ref.get_and_reset_other_loss()['reg_loss'].sum()
            for p in params.values():
                p.grad = None
            loss.backward()
            for k, p in params.items():
                if p.grad is not None:
                    grads[k] += p.grad / 2
            close(got[r]['losses'][s], float(losses_ref.bpr_loss(logits, labels).detach()), what=f'rank {r} loss {s}', rtol=2e-4, atol=1e-5)
        sc = gscale(grads.values())
        for k, p in params.items():
            p.grad = grads[k]
            # the averaged gradient both replicas apply in this step (norm-wise, like the single-GPU gradient tests: a shard of
            # three rows makes the BatchNorm backward ill-conditioned, so single elements carry per cents of relative noise)
            for r in range(2):
                close(got[r][f'g{s}/{k}'], grads[k], what=f'step {s} rank {r} averaged gradient {k}', rtol=2e-4, atol=1e-6, scale=sc,
                      norm_rtol=2e-4)
        opt.step()
    skip = bn_shadowed_biases(sd.keys())
    for k, v in params.items():
        assert np.array_equal(got[0][k], got[1][k]), f'replicas differ in {k}'
        if k not in skip:
            # Adam / Adagrad normalise each element's step by its own gradient history: where consecutive gradients nearly
            # cancel, the per cents of noise above become per cents of lr. 2 % of the distance three steps can move an element
            # is granted on top of the G8 tolerance (a wrong average would be off by the whole distance)
            close(got[0][k], v.detach(), what=k, rtol=2e-4, atol=2e-5 + 0.02 * 3 * case['lr'], norm_rtol=1e-4)
    for r in range(2):
        for k, v in stats[r].items():
            if k not in skip:
                close(got[r][k], v, what=f'rank {r} {k}', rtol=2e-4, atol=2e-5, norm_rtol=1e-4)


def test_two_rank_item_sharded_evaluation_equals_one_rank(tmp_path):
    """``evaluate_recommender_algorithm`` under a 2-rank process group (item-sharded: eval/eval.py:203-222 with BASELINE configs[4]'s
    sharding — each rank the representations and scores of its item shard, all-gather + exact merge of the top-k lists, metrics on
    every rank) returns on BOTH ranks, user by user, what the one-rank evaluation returns: the golden G9 world (fp32 route) and a
    3,000 x 1,111 world with D = 64 (fused kernel with item_offset), both scorers. See tests/dp_eval_worker.py."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR')}
    prefix = str(tmp_path / 'ev')
    def launch(world):
        procs = [subprocess.Popen([sys.executable, os.path.join(here, 'dp_eval_worker.py'), str(r), str(world), str(tmp_path / f'rdzv{world}'), prefix],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        logs = [p.communicate(timeout=600)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), '\n'.join(l[-2000:] for l in logs)
        return [np.load(prefix + f'.w{world}.rank{r}.npz') for r in range(world)]
    one = launch(1)[0]
    two = launch(2)
    assert len(one.files) > 20
    for r in range(2):
        assert sorted(two[r].files) == sorted(one.files)
        for k in one.files:
            assert np.array_equal(two[r][k], one[k]), f'rank {r}: {k} differs from the one-rank evaluation'

