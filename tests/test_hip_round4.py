"""GPU: round-4 items.

  a) lab switches are not in the product library: SBR_ST_DEBUG / SBR_ST_PRE in the environment change nothing;
  b) a lookup user table updated row by row (engine.DeferredTable) survives load_state_dict() and a change of lr / wd
     (loaded weights must not take the zero-gradient steps the OLD rows still owed);
  c) the fused scorer on an item shard SHORTER than the list (k > items): empty slots (-inf, -1) behind the shard's items
     (eval/eval.py:216-222 item-sharded as BASELINE configs[4] asks);
  d) c4 (BASELINE configs[3]: 1M users x 200k items, D = 256) at full table shapes against the CPU oracle, then five deferred
     AdamW steps with the sweep period the engine picks, against torch.optim.AdamW (train/trainer.py:62-68).
"""
import os

import numpy as np
import pytest
import torch

from golden_util import close

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def S():
    import sibrar_amd
    return sibrar_amd


def _excl_csr(U, I, per, seed):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, I, size=(U, per))
    m = sp.csr_matrix((np.ones(U * per, dtype=np.int8), cols.reshape(-1), np.arange(0, U * per + 1, per)), shape=(U, I))
    m.sum_duplicates()
    return S().evaluation._csr_to_device(m, DEV)


def _ref_topk(u16, i16, k, users=None, ex=None, item_offset=0):
    sc = u16.float() @ i16.float().t()
    if ex is not None:
        S().ops.mask_scores_(sc, users, ex[0], ex[1], item_offset=item_offset if item_offset else None)
    return S().ops.topk_rows(sc, min(k, i16.shape[0]))


# ---- a) ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('D', [64, 128, 256])
def test_lab_environment_switches_do_not_reach_the_product_scorer(D, monkeypatch):
    """SBR_ST_DEBUG selected timing-only ablations of the scorer (garbage results by design) in round 3's product library. They are
    compiled only into lab builds now: with the variables set the product returns the exact top-k."""
    g = torch.Generator().manual_seed(D)
    U, I, k = 700, 9000, 20
    u16 = (torch.randn(U, D, generator=g) / 4).half().to(DEV)
    i16 = (torch.randn(I, D, generator=g) / 4).half().to(DEV)
    ex = _excl_csr(U, I, 9, D)
    users = torch.arange(U, device=DEV)
    rv, ri = _ref_topk(u16, i16, k, users, ex)
    for dbg in ('1', '2', '5', '8'):
        monkeypatch.setenv('SBR_ST_DEBUG', dbg)
        monkeypatch.setenv('SBR_ST_PRE', '0')
        val, idx = S().ops.score_topk_f16(u16, i16, k, users, ex[0], ex[1])
        assert torch.equal(idx.long(), ri.long()), f'SBR_ST_DEBUG={dbg} changed the product scorer\'s top-k'
        close(val, rv, rtol=1e-5, atol=1e-6, what='scores')


# ---- b) ------------------------------------------------------------------------------------------------------------------------
_CFG = {'shared_common_dim': 32, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
        'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'item_embedding'}],
                 'single_branch_hidden_layers': [32], 'preference_hidden_layers': [], 'common_modality_dim': 32}}
_KEY = 'user_embedding_module.embedding_layer.weight'


def _world():
    ds = S().SyntheticDataset(2000, 200, 9000, item_dense={'text': 40}, seed=3, n_negative_samples=3, holdout_per_user=1)
    return ds


def _batch(rng, ds, s_):
    u = torch.from_numpy(rng.permutation(60 if s_ % 5 == 0 else ds.n_users)[:48].copy())
    i = torch.from_numpy(rng.integers(0, ds.n_items, size=(48, 4)))
    labels = torch.zeros(48, 4, dtype=torch.float64)
    labels[:, 0] = 1
    return u, i, labels


def test_load_state_dict_under_a_deferred_table_keeps_the_checkpoint(monkeypatch, tmp_path):
    """Train with the deferred row-wise AdamW (rows lag behind), save, train on, load the checkpoint (load_model_from_path ->
    load_state_dict), evaluate (state_dict / forward flush the table): the user table must be bit-equal to the checkpoint — the
    zero-gradient steps the rows owed BEFORE the load belong to the old weights and must not be replayed on the loaded ones."""
    monkeypatch.setenv('SBR_DEFERRED_ADAM', '1')
    ds = _world()
    torch.manual_seed(11)
    np.random.seed(11)
    net = S().SingleBranchNet(S().SingleBranchNetConfig.from_dict(_CFG), ds).to(DEV)
    net.train()
    opt = S().FusedOptimizer(net, 'adamw', lr=1e-2, weight_decay=1e-2)
    loss = S().RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
    fused = S().FusedTrainStep(net, loss, opt)
    assert fused.deferred is not None
    rng = np.random.default_rng(9)
    for s_ in range(8):
        fused.step(*_batch(rng, ds, s_))
    net.save_model_to_path(str(tmp_path))
    ckpt = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for s_ in range(8, 20):                                        # rows are behind again when the checkpoint comes back
        fused.step(*_batch(rng, ds, s_))
    assert int((fused.deferred.last < opt.step_count).sum()) > 0, 'the test needs rows that still owe steps'
    net.load_model_from_path(str(tmp_path))
    ev = ds.eval_view()
    evaluator = S().FullEvaluator(config=S().evaluation._Cfg(top_k=(10,), metrics=['ndcg'], calculate_std=False), dataset=ev)
    S().evaluate_recommender_algorithm(net, type('L', (), {'dataset': ev, 'batch_size': 256})(), evaluator, DEV)
    now = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for k, v in ckpt.items():
        assert torch.equal(now[k], v), f'{k} differs from the loaded checkpoint'
    # and training continues from the loaded weights: one more step changes only what a step can change
    net.train()
    fused.step(*_batch(rng, ds, 20))
    after = net.state_dict()[_KEY].detach().cpu()
    moved = (after - ckpt[_KEY]).abs().max().item()
    assert 0 < moved < 0.2, f'one step after the load moved the user table by {moved}'
    fused.close()


@pytest.mark.parametrize('what', ['lr', 'wd'])
def test_changing_lr_or_wd_under_a_deferred_table_equals_the_dense_optimizer(what, monkeypatch):
    """A row's replayed zero-gradient steps use the hyper-parameters of the call that replays them; the optimizer therefore brings the
    table up to date before lr / wd change. 12 steps, a change after step 5: the deferred run equals the dense run."""
    runs = []
    for deferred in ('0', '1'):
        monkeypatch.setenv('SBR_DEFERRED_ADAM', deferred)
        ds = _world()
        torch.manual_seed(11)
        np.random.seed(11)
        net = S().SingleBranchNet(S().SingleBranchNetConfig.from_dict(_CFG), ds).to(DEV)
        net.train()
        opt = S().FusedOptimizer(net, 'adamw', lr=1e-2, weight_decay=1e-2)
        loss = S().RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
        fused = S().FusedTrainStep(net, loss, opt)
        rng = np.random.default_rng(9)
        for s_ in range(12):
            if s_ == 5:
                if what == 'lr':
                    opt.lr = 2e-3
                else:
                    opt.wd = 0.2
            u, i, labels = _batch(rng, ds, s_)
            u = torch.unique(u)[:40]                               # distinct users: no atomics-order noise in the table gradient
            fused.step(u, i[:len(u)], labels[:len(u)])
        fused.close()
        runs.append(net.state_dict()[_KEY].detach().cpu().clone())
    close(runs[1], runs[0], what=f'user table after a change of {what}', rtol=1e-5, atol=1e-6)


# ---- c) ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('D,I,k,offset', [(64, 5, 10, 0), (128, 7, 20, 1000), (256, 19, 20, 64), (128, 1, 3, 0)])
def test_fused_scorer_on_a_shard_shorter_than_the_list(D, I, k, offset):
    """An item shard with fewer items than the list is long (world > n_split / k in an item-sharded evaluation): the shard's items in
    order, then empty slots (-inf, -1); exclusions inside the shard honoured, item_offset added to the positions."""
    g = torch.Generator().manual_seed(I)
    U = 300
    u16 = (torch.randn(U, D, generator=g) / 4).half().to(DEV)
    i16 = (torch.randn(I, D, generator=g) / 4).half().to(DEV)
    # exclusion CSR over the WHOLE catalogue (offset + I + 50 items): some entries fall into the shard
    ex = _excl_csr(U, offset + I + 50, 6, I)
    users = torch.arange(U, device=DEV)
    val, idx = S().ops.score_topk_f16(u16, i16, k, users, ex[0], ex[1], item_offset=offset)
    sc = u16.float() @ i16.float().t()
    S().ops.mask_scores_(sc, users, ex[0], ex[1], item_offset=offset)       # the shard-aware mask: CSR columns outside the shard are skipped
    sc_c, val_c, idx_c = sc.cpu(), val.cpu(), idx.cpu().long()
    for u in range(U):
        order = sorted(range(I), key=lambda j: (-float(sc_c[u, j]), j))
        order = [j for j in order if sc_c[u, j] != -float('inf')]
        n = min(len(order), k)
        assert idx_c[u, :n].tolist() == [j + offset for j in order[:n]], f'user {u}'
        assert (idx_c[u, n:] == -1).all() and (val_c[u, n:] == -float('inf')).all(), f'user {u}: empty slots'


# ---- d) ------------------------------------------------------------------------------------------------------------------------
_C4_CFG = {'shared_common_dim': 256, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'image'}], 'single_branch_hidden_layers': [256],
                    'preference_hidden_layers': [], 'common_modality_dim': 256}}


def _c4_world():
    ds = S().SyntheticDataset(1_000_000, 200_000, 4_000_000, item_dense={'text': 768, 'image': 2048}, seed=0, n_negative_samples=10)
    torch.manual_seed(42)
    np.random.seed(42)
    net = S().SingleBranchNet(S().SingleBranchNetConfig.from_dict(_C4_CFG), ds).to(DEV).train()
    assert sum(p.numel() for p in net.parameters()) > 256_000_000
    return ds, net


def test_c4_step_at_full_table_shapes_against_the_cpu_oracle(monkeypatch):
    """BASELINE configs[3] on one GPU (what every rank of the 8-GPU job runs): 1M users x 200k items, text 768 + image 2048, C = D =
    256, user = embedding lookup (1 GB table), sampled softmax, ONE batch-256 step of ``FusedTrainStep`` — plain launches, then capture +
    replay — against the CPU oracle (oracle/model_ref.py restating train/trainer.py:204-223 -> sgd_alg.py:2116-2125,
    rec_losses.py:88-113) on the same parameters, batch and modality draw: the loss (1e-4 relative) and EVERY gradient, incl. the rows
    of the 1 GB user table the batch touches and the sum of |gradient| over all of its rows (untouched rows: exact zeros)."""
    from golden_util import gscale
    from oracle import losses_ref, model_ref
    monkeypatch.setenv('SBR_DEFERRED_ADAM', '1')
    ds, net = _c4_world()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and 'running' not in k:
            v.requires_grad_(True)
    opt = S().FusedOptimizer(net, 'adamw', lr=1e-3, weight_decay=0.)
    lossf = S().RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
    fused = S().FusedTrainStep(net, lossf, opt, use_graph=True)
    seen = []
    def _record(*a, **k):                                      # stands in for the optimizer launch (which also resets the gradient)
        seen.append({k_: p.grad.detach().clone() for k_, p in net.named_parameters()})
        if k.get('zero_grad'):
            opt.fp.grad.zero_()
        return False
    opt.step_flat = _record
    rng = np.random.default_rng(8)
    u = torch.from_numpy(rng.integers(0, ds.n_users, size=256))
    i = torch.from_numpy(rng.integers(0, ds.n_items, size=(256, 11)))
    labels = torch.zeros(256, 11, dtype=torch.float64)
    labels[:, 0] = 1
    draws = fused.draw(u.shape, i.shape)
    recs = []
    for rep in range(5):
        total, rec, reg = fused.step(u, i, labels, draws)
        recs.append(rec.cpu())
    assert fused.n_replays >= 1
    pos, order = draws[1]
    mods = np.array(order)[pos].reshape(tuple(i.shape) + (1,))
    ut = {'user_embedding': model_ref.RefTable('categorical', np.arange(ds.n_users), n_categories=ds.n_users)}
    it = {k: model_ref.table_from_feature(f) for k, f in ds.item_features.items()}
    ref = model_ref.RefSingleBranchNet(sd, _C4_CFG, ut, it, orders={'item_train': net.item_embedding_module.train_modality_order,
                                                                   'item_eval': net.item_embedding_module.eval_modality_order})
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    logits = ref.forward(u, i, True, None, mods)
    rl = losses_ref.RefRecLoss('sampled_softmax', n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole',
                               neg_train=10).compute_loss(logits, labels)
    rl.backward()
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    key = 'user_embedding_module.embedding_layer.weight'
    assert key in grads and len(grads) == len(seen[0])
    sc = gscale(grads.values())
    rows = torch.unique(u)
    for rep in (0, len(seen) - 1):                              # the first plain pass and the last replay
        close(recs[rep], rl.detach().double(), what=f'rec loss (pass {rep})', rtol=1e-4, atol=1e-7)
        for k_, g in grads.items():
            got = seen[rep][k_]
            if k_ == key:
                close(got[rows.to(DEV)].cpu(), g[rows], what=f'grad of the touched user rows (pass {rep})', rtol=2e-4, atol=1e-7, scale=sc, norm_rtol=1e-4)
                tot_got, tot_ref = float(got.abs().double().sum()), float(g.abs().double().sum())
                assert abs(tot_got - tot_ref) <= 1e-4 * tot_ref, f'sum |grad| of the user table: {tot_got} vs {tot_ref}'
                assert float(got.abs().double().sum() - got[rows.to(DEV)].abs().double().sum()) == 0.0, 'untouched rows carry gradient'
            else:
                close(got.cpu(), g, what=f'grad {k_} (pass {rep})', rtol=2e-4, atol=1e-7, scale=sc, norm_rtol=1e-4)
    fused.close()


def test_c4_five_deferred_steps_with_the_engines_sweep_against_torch_adamw(monkeypatch):
    """The same world, five REAL steps with the deferred row-wise AdamW and the sweep period ``engine.DeferredTable`` picks for this table
    (768: 1 / 768 of the 1M rows per step), batches drawn from a pool of 700 users so that rows are touched, left alone for a step or
    three and touched again (their catch-up replays the missed zero-gradient steps), then ``flush()``: the 1 GB user table against
    torch.optim.AdamW (oracle/train_ref.make_optimizer, train/trainer.py:62-68) fed with the same per-step table gradients — every
    touched row, a stride sample of never-touched rows (five steps of pure decoupled weight decay) and two checksums of the whole
    table."""
    from oracle import train_ref
    monkeypatch.setenv('SBR_DEFERRED_ADAM', '1')
    ds, net = _c4_world()
    lr, wd = 1e-2, 1e-2
    opt = S().FusedOptimizer(net, 'adamw', lr=lr, weight_decay=wd)
    lossf = S().RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
    fused = S().FusedTrainStep(net, lossf, opt)
    assert fused.deferred is not None and fused.deferred.sweep_period() == 768
    table = net.user_embedding_module.embedding_layer.weight
    p_ref = torch.nn.Parameter(table.detach().cpu().clone())
    ref_opt = train_ref.make_optimizer('adamw', [p_ref], lr, wd)
    real_rows_step = fused.deferred.step
    grads = []
    def _step(ids, copy=None):                                  # the table gradient of this step, read before the launch consumes it
        grads.append(table.grad.detach().cpu().clone())
        return real_rows_step(ids, copy=copy)
    fused.deferred.step = _step
    rng = np.random.default_rng(12)
    pool = rng.choice(ds.n_users, size=700, replace=False)
    touched = set()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for s_ in range(5):
        u = torch.from_numpy(rng.choice(pool, size=256, replace=False))
        i = torch.from_numpy(rng.integers(0, ds.n_items, size=(256, 11)))
        labels = torch.zeros(256, 11, dtype=torch.float64)
        labels[:, 0] = 1
        total, rec, reg = fused.step(u, i, labels)
        assert torch.isfinite(total).all()
        touched.update(u.tolist())
        assert len(grads) == s_ + 1, 'the optimizer launch of the step did not go through DeferredTable.step'
        p_ref.grad = grads[-1]
        ref_opt.step()
    assert int((fused.deferred.last < opt.step_count).sum()) > 0, 'rows must be behind before the flush'
    fused.flush()
    rows = torch.tensor(sorted(touched) + list(range(0, 1_000_000, 9973)))
    got = table.detach()
    close(got[rows.to(DEV)].cpu(), p_ref.detach()[rows], what='touched + sampled rows after five deferred steps', rtol=1e-5, atol=1e-7)
    want = p_ref.detach().double()
    assert abs(float(got.double().sum()) - float(want.sum())) <= 1e-6 * float(want.abs().sum())
    assert abs(float(got.double().pow(2).sum()) - float(want.pow(2).sum())) <= 1e-6 * float(want.pow(2).sum())
    fused.close()


# ---- e) BASELINE configs[4] end to end on one GPU ---------------------------------------------------------------------------------
def test_c5_eight_item_shards_merged_equal_the_unsharded_pass():
    """What the 8-GPU job of BASELINE configs[4] computes, rank by rank on one GPU: 200k items x 256 fp16 cut into eight shards of 25k
    (parallel.item_shard), every shard scored with the fused kernel at its item_offset against the same users (a 20k-user chunk, 50
    exclusions per user over the whole catalogue), the eight [U, 20] lists stacked as an all-gather would leave them and merged by
    sbr_merge_topk — against ONE unsharded pass over the whole catalogue: the same lists bit for bit (both scorer routes)."""
    from importlib import import_module
    _lib = import_module(S().ops.__name__.rsplit('.', 1)[0] + '._lib')
    g = torch.Generator().manual_seed(5)
    U, I, D, k, W = 20_000, 200_000, 256, 20, 8
    u16 = (torch.randn(U, D, generator=g) / 16).half().to(DEV)
    i16 = (torch.randn(I, D, generator=g) / 16).half().to(DEV)
    ex = _excl_csr(U, I, 50, 5)
    users = torch.arange(U, device=DEV)
    for route in (1, 2):
        prev = S().ops.score_topk_route(route)
        try:
            full_v, full_i = S().ops.score_topk_f16(u16, i16, k, users, ex[0], ex[1])
            vals = torch.empty(W, U, k, device=DEV, dtype=torch.float32)
            idxs = torch.empty(W, U, k, device=DEV, dtype=torch.int32)
            for r in range(W):
                lo, hi = S().parallel.item_shard(I, r, W)
                assert hi - lo == 25_000
                vals[r], idxs[r] = S().ops.score_topk_f16(u16, i16[lo:hi].contiguous(), k, users, ex[0], ex[1], item_offset=lo)
        finally:
            S().ops.score_topk_route(prev)
        out_v, out_i = torch.empty(U, k, device=DEV), torch.empty(U, k, device=DEV, dtype=torch.int32)
        _lib.call('sbr_merge_topk', vals.data_ptr(), idxs.data_ptr(), W, U, k, out_v.data_ptr(), out_i.data_ptr(), _lib.stream())
        assert torch.equal(out_i, full_i) and torch.equal(out_v, full_v), f'route {route}: sharded + merged differs from the unsharded pass'
        assert bool((full_v[:, :-1] >= full_v[:, 1:]).all()) and int(full_i.min()) >= 0 and int(full_i.max()) < I
