"""GPU: the HIP engine behind the reference's plugin surface against (a) the golden vectors captured from the real
reference and (b) the CPU oracle on the same inputs. Tolerance: fp32 logits / losses within 1e-4 relative (north star);
gradients within 1e-4 of the gradient scale (see golden_util.close); integer work exact."""
import numpy as np
import pytest
import torch

from golden_util import (MANIFEST, load, sub, state_dict, world, ref_tables, close, gscale, bn_shadowed_biases, host_dataset,
                         product_net, side_cfg, U, I)

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = dict(rtol=1e-4, atol=1e-5)


def _grads_close(module_or_params, golden, prefix=''):
    gs = golden
    sc = gscale(gs.values())
    named = dict(module_or_params.named_parameters()) if hasattr(module_or_params, 'named_parameters') else module_or_params
    for k, g in gs.items():
        p = named[k]
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        close(got.detach().cpu(), g, what=f'grad {k}', rtol=2e-4, atol=1e-5, scale=sc, norm_rtol=1e-4)


@pytest.mark.parametrize('case', MANIFEST['g1_polylinear']['cases'], ids=lambda c: c['name'])
def test_g1_polylinear(case):
    import sibrar_amd as S
    z = load('g1_polylinear')
    n = case['name']
    pl = S.PolyLinear(case['layer_config'], activation_fn=case['act'], output_fn=case['out_act'],
                      apply_batch_norm_every=case['bn_every'])
    pl.load_state_dict(state_dict(z, f'{n}/sd0/'), strict=True)
    pl.to(DEV).train()
    for s in range(3):
        x = torch.from_numpy(z[f'{n}/x{s}']).to(DEV).requires_grad_(True)
        y = pl(x)
        close(y.detach().cpu(), z[f'{n}/y{s}'], what=f'y{s}', **TOL)
        pl.zero_grad()
        (y * torch.from_numpy(z[f'{n}/r{s}']).to(DEV)).sum().backward()
        gs = sub(z, f'{n}/g{s}/')
        close(x.grad.cpu(), z[f'{n}/gx{s}'], what=f'gx{s}', rtol=2e-4, atol=1e-5, scale=gscale(gs.values()), norm_rtol=1e-4)
        _grads_close(pl, gs)
    for k, v in sub(z, f'{n}/sd3/').items():
        close(pl.state_dict()[k].cpu(), v, what=f'sd3/{k}', **TOL)
    pl.eval()
    with torch.no_grad():
        ye = pl(torch.from_numpy(z[f'{n}/xe']).to(DEV))
    close(ye.cpu(), z[f'{n}/ye'], what='ye', **TOL)


def _g2_feature(w, source):
    import sibrar_amd as S
    ds = host_dataset(w)
    if source == 'item_interactions':
        return S.HostFeature('interactions', 'csr', w['inter_t'])
    if source == 'gender':
        return ds.user_features['gender']
    return ds.item_features[source]


@pytest.mark.parametrize('case', MANIFEST['g2_feature_embedding']['cases'], ids=lambda c: c['name'])
def test_g2_feature_embedding(case):
    import sibrar_amd as S
    z = load('g2_feature_embedding')
    n = case['name']
    fe = S.FeatureEmbedding(_g2_feature(world(z), case['source']), embedding_dim=case['embedding_dim'],
                            pre_embedding_layers=case['hidden'], activation_fn=case['act'])
    fe.load_state_dict(state_dict(z, f'{n}/sd/'), strict=True)
    fe.to(DEV).train()
    idx = torch.from_numpy(z[f'{n}/idx']).to(DEV)
    y = fe(idx)
    y = y.reshape(z[f'{n}/y'].shape)
    close(y.detach().cpu(), z[f'{n}/y'], what='y', **TOL)
    (y * torch.from_numpy(z[f'{n}/r']).to(DEV)).sum().backward()
    _grads_close(fe, sub(z, f'{n}/g/'))


@pytest.mark.parametrize('case', MANIFEST['g3_entity']['cases'], ids=lambda c: c['name'])
def test_g3_entity(case):
    import sibrar_amd as S
    z = load('g3_entity')
    n = case['name']
    w = world(z)
    ds = host_dataset(w)
    feats = dict(ds.item_features)
    feats['interactions'] = S.HostFeature('interactions', 'csr', w['inter_t'])
    cfg = S.SingleBranchNetEntityConfig.from_dict(side_cfg(case['cfg']))
    order = case['train_order']
    if case.get('central_others_order'):
        order = [case['cfg']['central_modality']] + case['central_others_order']
    ent = S.SingleBranchNetEntity('item', feats, cfg, 8, True, train_modality_order=order,
                                  eval_modality_order=case['eval_order'])
    ent.load_state_dict(state_dict(z, f'{n}/sd0/'), strict=True)
    ent.to(DEV).train()
    idx = torch.from_numpy(z[f'{n}/idx']).to(DEV)
    # (1) own draw: same stream as the reference's first call
    pos, used_order = ent._sample_modalities(tuple(idx.shape))
    assert (ent.modality_names(pos, used_order).reshape(z[f'{n}/mods'].shape) == z[f'{n}/mods']).all()
    # (2) replay the recorded decision
    y = ent(idx, modalities=z[f'{n}/mods'])
    close(y.detach().cpu(), z[f'{n}/y'], what='y', **TOL)
    reg = ent.get_and_reset_other_loss()['reg_loss']
    close(reg.detach().cpu().reshape(-1), z[f'{n}/reg_loss'].reshape(-1), what='reg', **TOL)
    ((y * torch.from_numpy(z[f'{n}/r']).to(DEV)).sum() + reg.sum()).backward()
    _grads_close(ent, sub(z, f'{n}/g/'))
    for k, v in sub(z, f'{n}/sd1/').items():
        close(ent.state_dict()[k].cpu(), v, what=f'sd1/{k}', **TOL)
    ent.eval()
    with torch.no_grad():
        ye = ent(torch.arange(I, device=DEV))
    close(ye.cpu(), z[f'{n}/y_eval'], what='y_eval', **TOL)


_LOSS = {
    'bce': ('bce', 'mean', 'uniform_recbole'), 'bpr': ('bpr', 'mean', 'uniform_recbole'), 'bpr_sum': ('bpr', 'sum', 'uniform_recbole'),
    'ssm_uniform': ('sampled_softmax', 'mean', 'uniform'), 'ssm_recbole': ('sampled_softmax', 'sum', 'uniform_recbole'),
}


def _loss(name):
    import sibrar_amd as S
    kind, agg, strat = _LOSS[name]
    return S.RecommenderSystemLossesEnum[kind].value(n_items=I, aggregator=agg, train_neg_strategy=strat, neg_train=3)


@pytest.mark.parametrize('case', MANIFEST['g4_full_net']['cases'], ids=lambda c: c['name'])
def test_g4_full_net(case):
    z = load('g4_full_net')
    n = case['name']
    net = product_net(z, case, f'{n}/sd0/')
    net.train()
    u, i, labels = (torch.from_numpy(z[f'{n}/{k}']).to(DEV) for k in ('u', 'i', 'labels'))
    um = z[f'{n}/user_mods'] if f'{n}/user_mods' in z.files else None
    logits = net(u, i, user_modalities=um, item_modalities=z[f'{n}/item_mods'])
    assert logits.dtype == torch.float32
    close(logits.detach().cpu(), z[f'{n}/logits'], what='logits', **TOL)
    loss = _loss(case['loss']).compute_loss(logits, labels)
    assert str(loss.dtype) == case['rec_loss_dtype']
    close(loss.detach().cpu(), z[f'{n}/rec_loss'], what='rec_loss', **TOL)
    reg = net.get_and_reset_other_loss()
    close(reg['reg_loss'].detach().cpu(), z[f'{n}/reg_loss'], what='reg_loss', **TOL)
    (loss + reg['reg_loss']).backward()
    _grads_close(net, sub(z, f'{n}/g/'))
    for k, v in sub(z, f'{n}/sd1/').items():
        close(net.state_dict()[k].cpu(), v, what=f'sd1/{k}', **TOL)


@pytest.mark.parametrize('case', MANIFEST['g5_infonce']['cases'], ids=lambda c: c['name'])
def test_g5_infonce(case):
    import sibrar_amd as S
    z = load('g5_infonce')
    n = case['name']
    a = torch.from_numpy(z[f'{n}/a']).to(DEV).requires_grad_(True)
    b = torch.from_numpy(z[f'{n}/b']).to(DEV).requires_grad_(True)
    loss = S.InfoNCE(case['tau'], case['reduction'])(a, b)
    close(loss.detach().cpu(), z[f'{n}/loss'], what='loss', **TOL)
    loss.backward()
    close(a.grad.cpu(), z[f'{n}/ga'], what='ga', rtol=2e-4, atol=1e-5, norm_rtol=1e-4)
    close(b.grad.cpu(), z[f'{n}/gb'], what='gb', rtol=2e-4, atol=1e-5, norm_rtol=1e-4)


@pytest.mark.parametrize('case', MANIFEST['g8_optim']['cases'], ids=lambda c: c['name'])
def test_g8_optimizer_trajectories(case):
    """Three full training steps (forward, loss, backward, fused dense optimizer) replaying the reference's batches and
    modality decisions; every parameter must end where the reference's ended (BN-shadowed biases excluded, see
    golden_util.bn_shadowed_biases)."""
    import sibrar_amd as S
    z = load('g8_optim')
    n = case['name']
    net = product_net(z, case, f'{n}/sd0/')
    net.train()
    opt = S.FusedOptimizer(net, case['optimizer'], lr=case['lr'], weight_decay=case['wd'])
    loss_fn = S.RecBayesianPersonalizedRankingLoss(n_items=I, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
    for s in range(3):
        u, i, labels = (torch.from_numpy(z[f'{n}/{k}{s}']).to(DEV) for k in ('u', 'i', 'labels'))
        um = z[f'{n}/user_mods{s}'] if f'{n}/user_mods{s}' in z.files else None
        logits = net(u, i, user_modalities=um, item_modalities=z[f'{n}/item_mods{s}'])
        loss = loss_fn.compute_loss(logits, labels)
        reg = net.get_and_reset_other_loss()
        close(loss.detach().cpu(), z[f'{n}/loss{s}'], what=f'loss{s}', rtol=2e-4, atol=1e-5)
        (loss + reg['reg_loss']).backward()
        opt.step()
        opt.zero_grad()
    final = sub(z, f'{n}/sd3/')
    skip = bn_shadowed_biases(final.keys())
    sd = net.state_dict()
    for k, v in final.items():
        if k not in skip:
            close(sd[k].cpu(), v, what=f'sd3/{k}', rtol=2e-4, atol=2e-5, norm_rtol=1e-4)


def test_g9_eval_fp32_path():
    """Item representations, all-pairs scores, CSR mask, exact top-k and the ranking metrics against the reference's."""
    import sibrar_amd as S
    z = load('g9_eval')
    case = MANIFEST['g9_eval']
    net = product_net(z, case, 'sd/')
    net.eval()
    w = world(z)
    with torch.no_grad():
        i_repr = net.get_item_representations(torch.arange(I, device=DEV))
        u_idx = torch.arange(U, device=DEV)
        u_repr = net.get_user_representations(u_idx)
        close(i_repr.cpu(), z['i_repr'], what='i_repr', **TOL)
        close(u_repr.cpu(), z['u_repr'], what='u_repr', **TOL)
        out = net.combine_user_item_representations(u_repr, i_repr)
        indptr = torch.from_numpy(w['inter'].indptr.astype(np.int64)).to(DEV)
        indices = torch.from_numpy(w['inter'].indices.astype(np.int32)).to(DEV)
        S.ops.mask_scores_(out, u_idx, indptr, indices)
        ref = torch.from_numpy(z['scores'])
        assert (torch.isinf(out.cpu()) == torch.isinf(ref)).all()
        fin = ~torch.isinf(ref)
        close(out.cpu()[fin], ref[fin], what='scores', **TOL)
        val, idx = S.ops.topk_rows(out, 20)
    # top-k values equal the reference's (ranks may differ only among exact ties)
    close(torch.nan_to_num(val.cpu(), neginf=-1e30), np.nan_to_num(z['topk_val'], neginf=-1e30), what='topk_val', **TOL)
    # exactness of the selection itself on the engine's own scores
    tv, ti = torch.topk(out, 20, sorted=True)
    assert torch.equal(val, tv)
    labels = z['labels']
    import scipy.sparse as sp
    lab = sp.csr_matrix(labels)
    m = S.ops.rank_metrics(idx, u_idx, torch.from_numpy(lab.indptr.astype(np.int64)).to(DEV),
                           torch.from_numpy(lab.indices.astype(np.int32)).to(DEV), [1, 10, 20])
    for qi, k in enumerate([1, 10, 20]):
        close(m[0, qi].cpu(), z[f'ndcg@{k}'], what=f'ndcg@{k}', rtol=1e-5, atol=1e-6)
        close(m[1, qi].cpu(), z[f'recall@{k}'], what=f'recall@{k}', rtol=1e-5, atol=1e-6)
        close(m[2, qi].cpu(), z[f'precision@{k}'], what=f'precision@{k}', rtol=1e-5, atol=1e-6)


def test_g11_sgd_baseline():
    import sibrar_amd as S
    z = load('g11_sgd_baseline')
    m = S.SGDBaseline(U, I)
    m.load_state_dict(state_dict(z, 'sd/'))
    m.to(DEV)
    u, i = torch.from_numpy(z['u']).to(DEV), torch.from_numpy(z['i']).to(DEV)
    close(m.predict(u, i).cpu(), z['logits'], what='logits', **TOL)
    m.train()
    close(m(u, i).detach().cpu(), z['logits'], what='logits(train)', **TOL)


def _g15_cases():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g15_dropoutnet.json')))['cases']


@pytest.mark.parametrize('case', _g15_cases(), ids=lambda c: c['name'])
def test_g15_dropoutnet_on_hip_kernels(case):
    """DropoutNet of the product (device-side dense preference vectors, PolyLinear MFMA GEMMs, FeatureEmbedding content modules,
    scorers) == the real reference: the preference-dropout draws, train-mode logits, BPR loss, every gradient, evaluation
    scores; config dictionaries parse through DropoutNetConfig.from_dict."""
    import sibrar_amd as S
    z = load('g15_dropoutnet')
    name = case['name']
    m = S.DropoutNet.build_from_conf(case['config'], host_dataset(world(z)))
    sd = state_dict(z, f'{name}/sd/')
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    m.to(DEV).train()
    u, i, labels = (torch.from_numpy(z[k]).to(DEV) for k in ('u', 'i', 'labels'))
    logits = m(u, i)                                            # draws its own strategies: must be the reference's stream
    close(logits.detach().cpu(), z[f'{name}/logits'], what='logits', **TOL)
    loss = _loss('bpr').compute_loss(logits, labels)
    close(loss.detach().cpu(), z[f'{name}/loss'], what='loss', **TOL)
    loss.backward()
    for k, p in m.named_parameters():
        g = p.grad.cpu() if p.grad is not None else torch.zeros(tuple(p.shape))
        close(g, z[f'{name}/grad/{k}'], what=f'grad {k}', rtol=1e-4, atol=1e-6, norm_rtol=1e-4)
    m.eval()
    with torch.no_grad():
        ir = m.get_item_representations(torch.arange(I, device=DEV))
        scores = m.combine_user_item_representations(m.get_user_representations(u), ir)
    close(scores.cpu(), z[f'{name}/scores_all'], what='all-pairs scores', **TOL)


def _g14_cases():
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g14_sibling_models.json')
    return json.load(open(here))['cases']


@pytest.mark.parametrize('case', [c for c in _g14_cases() if 'class' in c], ids=lambda c: c['name'])
def test_g14_hybrid_factorisation_models_on_hip_kernels(case):
    """ItemFeature / UserFeatureMatrixFactorization of the product (lookups, FeatureEmbedding front end, InfoNCE kernels, the
    modality-mean kernel, MF scorer + bias kernels) == the real reference: logits, contrastive loss, total loss, every gradient,
    evaluation scores through get_*_representations + combine."""
    import sibrar_amd as S
    z = load('g14_sibling_models')
    name = case['name']
    ds = host_dataset(world(z))
    m = getattr(S, case['class'])(ds, **case['kwargs'])
    sd = state_dict(z, f'{name}/sd/')
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    m.to(DEV).train()
    u, i, labels = (torch.from_numpy(z[k]).to(DEV) for k in ('u', 'i', 'labels'))
    logits = m(u, i)
    reg = m.get_and_reset_other_loss()['reg_loss']
    close(logits.detach().cpu(), z[f'{name}/logits'], what='logits', **TOL)
    close(torch.as_tensor(reg).detach().float().cpu().reshape(-1), z[f'{name}/reg_loss'], what='reg loss', **TOL)
    loss = _loss('bpr').compute_loss(logits, labels) + reg
    close(loss.detach().cpu().reshape(()), z[f'{name}/loss'], what='loss', **TOL)
    loss.backward()
    for k, p in m.named_parameters():
        g = p.grad.cpu() if p.grad is not None else torch.zeros(tuple(p.shape))
        close(g, z[f'{name}/grad/{k}'], what=f'grad {k}', rtol=1e-4, atol=1e-6, norm_rtol=1e-4)
    m.eval()
    with torch.no_grad():
        ir = m.get_item_representations(torch.arange(I, device=DEV))
        scores = m.combine_user_item_representations(m.get_user_representations(u), ir)
    close(scores.cpu(), z[f'{name}/scores_all'], what='all-pairs scores', **TOL)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [c for c in _g14_cases() if 'class' not in c] + [{'name': 'baseline'}], ids=lambda c: c['name'])
def test_g14_sibling_models_on_hip_kernels(case):
    """SGDMatrixFactorization / SGDBaseline of the product (lookups, per-slot dot / all-pairs MFMA GEMM, bias kernels and their
    hand-written backward) == the real reference: train-mode logits, BPR loss, every gradient, evaluation scores through
    get_*_representations + combine (eval/eval.py:209-217); 3 AdamW steps == the oracle with torch.optim on the CPU."""
    import sibrar_amd as S
    from oracle import model_ref, losses_ref, train_ref
    z = load('g14_sibling_models')
    name = case['name']
    if name == 'baseline':
        m, fn = S.SGDBaseline(U, I), model_ref.sgd_baseline_logits
    else:
        m = S.SGDMatrixFactorization(U, I, case['embedding_dim'], case['use_user_bias'], case['use_item_bias'], case['use_global_bias'])
        fn = model_ref.mf_logits
    sd = state_dict(z, f'{name}/sd/')
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    m.to(DEV).train()
    u, i, labels = (torch.from_numpy(z[k]).to(DEV) for k in ('u', 'i', 'labels'))
    logits = m(u, i)
    close(logits.detach().cpu(), z[f'{name}/logits'], what='logits', **TOL)
    loss = _loss('bpr').compute_loss(logits, labels)
    close(loss.detach().cpu(), z[f'{name}/loss'], what='loss', **TOL)
    loss.backward()
    for k, p in m.named_parameters():
        close(p.grad.cpu(), z[f'{name}/grad/{k}'], what=f'grad {k}', **TOL)
    m.eval()
    with torch.no_grad():
        ir = m.get_item_representations(torch.arange(I, device=DEV))
        scores = m.combine_user_item_representations(m.get_user_representations(u), ir)
    close(scores.cpu(), z[f'{name}/scores_all'], what='all-pairs scores', **TOL)
    # three optimizer steps against the oracle — with BCE: under BPR a global or user bias cancels in pos - neg, its gradient is
    # rounding noise and Adam turns that noise into +-lr steps (see golden_util.bn_shadowed_biases for the same effect)
    m.train()
    m.zero_grad()
    opt = S.FusedOptimizer(m, 'adamw', lr=1e-2, weight_decay=1e-2)
    ref_sd = state_dict(z, f'{name}/sd/', requires_grad=True)
    ref_opt = train_ref.make_optimizer('adamw', list(ref_sd.values()), 1e-2, 1e-2)
    ref_loss = losses_ref.RefRecLoss('bce', n_items=I, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
    rng = np.random.default_rng(3)
    for _ in range(3):
        bu = torch.from_numpy(rng.integers(0, U, size=16))
        bi = torch.from_numpy(rng.integers(0, I, size=(16, 4)))
        lab = torch.zeros(16, 4, dtype=torch.float64)
        lab[:, 0] = 1
        _loss('bce').compute_loss(m(bu.to(DEV), bi.to(DEV)), lab.to(DEV)).backward()
        opt.step()
        opt.zero_grad()
        ref_opt.zero_grad()
        ref_loss.compute_loss(fn(ref_sd, bu, bi), lab).backward()
        ref_opt.step()
    for k, v in m.state_dict().items():
        close(v.cpu(), ref_sd[k].detach(), what=f'{k} after 3 steps', rtol=1e-4, atol=1e-6, norm_rtol=1e-3)   # Adam: rounding of small gradients -> fractions of lr


def test_mf_user_bias_raises_like_the_reference():
    import sibrar_amd as S
    m = S.SGDMatrixFactorization(U, I, 8, use_user_bias=True).to(DEV)
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, dtype=torch.long, device=DEV), torch.zeros(4, 3, dtype=torch.long, device=DEV))


@pytest.mark.parametrize('user_kind,loss_name,B', [('lookup', 'bpr', 6), ('entity', 'ssm_uniform', 6), ('linear', 'bce', 6),
                                                   ('entity', 'bpr', 300)])
def test_fused_step_matches_autograd_path(user_kind, loss_name, B):
    """engine.FusedTrainStep (hand-written backward, no autograd) == module path (autograd over the same kernels): same
    losses and the same parameters after three AdamW steps on the golden world. B = 300 with an entity user side is the
    in-batch InfoNCE over 300 user rows (GEMM path of the loss; the on-chip kernel holds at most 176 rows)."""
    import sibrar_amd as S
    z = load('g4_full_net')
    case = [c for c in MANIFEST['g4_full_net']['cases'] if c['name'] == f'{user_kind}_bpr'][0]
    nets = [product_net(z, case, f"{case['name']}/sd0/") for _ in range(2)]
    for n_ in nets:
        n_.train()
    opts = [S.FusedOptimizer(n_, 'adamw', lr=1e-2, weight_decay=1e-2) for n_ in nets]
    fused = S.FusedTrainStep(nets[1], _loss(loss_name), opts[1])
    rng = np.random.default_rng(5)
    for s_ in range(3):
        u = torch.from_numpy(rng.integers(0, U, size=B))
        i = torch.from_numpy(rng.integers(0, I, size=(B, 4)))
        labels = torch.zeros(B, 4, dtype=torch.float64)
        labels[:, 0] = 1
        logits = nets[0](u.to(DEV), i.to(DEV))
        loss = _loss(loss_name).compute_loss(logits, labels.to(DEV))
        reg = nets[0].get_and_reset_other_loss()['reg_loss']
        (loss + reg).backward()
        opts[0].step()
        opts[0].zero_grad()
        total, rec, reg2 = fused.step(u, i, labels)
        close(rec.cpu(), loss.detach().cpu().double(), what=f'rec loss step {s_}', rtol=1e-5, atol=1e-7)
        close(reg2.cpu().reshape(-1), reg.detach().cpu().double().reshape(-1), what=f'reg loss step {s_}', rtol=1e-5, atol=1e-7)
    sd0, sd1 = nets[0].state_dict(), nets[1].state_dict()
    skip = bn_shadowed_biases(sd0.keys())
    for k in sd0:
        if k not in skip:
            close(sd1[k].cpu(), sd0[k].cpu(), what=k, rtol=1e-4, atol=1e-6, norm_rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('split', [False, True], ids=['one_graph', 'two_phase'])
def test_fused_step_graph_replay_equals_plain_launches(split, monkeypatch):
    """hipGraph replay of forward+backward (engine.FusedTrainStep, use_graph=True) == the same launches issued one by one:
    same losses and parameters after 12 AdamW steps on a small c2-shaped world (text modality + item-id embedding, both
    one of them drawn per index, so the per-modality row counts vary from step to step and the plans are padded to a
    bucket; a signature is captured at its second sighting). ``two_phase``: the launch structure of a data-parallel run (user
    side backward first, two graphs, the gradient exchange of the user part would start in between) on one GPU."""
    import sibrar_amd as S
    monkeypatch.setenv('SBR_FORCE_SPLIT', '1' if split else '0')
    ds = S.SyntheticDataset(300, 200, 6000, item_dense={'text': 40}, seed=3, n_negative_samples=3)
    cfg = {'shared_common_dim': 32, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'item_embedding'}],
                    'single_branch_hidden_layers': [32], 'preference_hidden_layers': [], 'common_modality_dim': 32}}
    runs = []
    for use_graph in (False, True):
        torch.manual_seed(11)
        np.random.seed(11)
        net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV)
        net.train()
        opt = S.FusedOptimizer(net, 'adamw', lr=1e-2, weight_decay=1e-2)
        loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
        fused = S.FusedTrainStep(net, loss, opt, use_graph=use_graph)
        rng = np.random.default_rng(9)
        losses = []
        for s_ in range(12):
            B = 64 if s_ != 4 else 48                     # one ragged batch in between (its own signature, plain launches)
            u = torch.from_numpy(rng.integers(0, ds.n_users, size=B))
            i = torch.from_numpy(rng.integers(0, ds.n_items, size=(B, 4)))
            labels = torch.zeros(B, 4, dtype=torch.float64)
            labels[:, 0] = 1
            losses.append(torch.stack(fused.step(u, i, labels)).cpu())
        runs.append((fused, losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
    assert runs[0][0].n_replays == 0 and runs[1][0].split == split
    assert runs[1][0].n_replays >= 4                        # 64 x 4 = 256 draws, bucket 64: at most 2-3 signatures
    for s_, (a, b) in enumerate(zip(runs[0][1], runs[1][1])):
        close(b, a, what=f'losses step {s_}', rtol=1e-6, atol=1e-9)
    # zero-gradient parameters (Adam turns their rounding noise into +-lr steps): biases in front of a BatchNorm, and the
    # trailing BatchNorm's shift — the softmax gradient sums to zero over each user's candidates, so a common shift of all
    # item representations has no gradient
    skip = set(bn_shadowed_biases(runs[0][2].keys())) | {'item_embedding_module.sb_net.1.bias'}
    for k in runs[0][2]:
        if k in skip:
            continue
        # not bit-equal: split-K chunking follows the (padded) row counts and the table gradients use float atomics; AdamW at
        # lr 1e-2 turns that rounding noise into ~1e-6 absolute differences on near-zero gradients
        close(runs[1][2][k].double(), runs[0][2][k].double(), what=k, rtol=1e-4, atol=1e-6, norm_rtol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('exchange', ['sparse_user_rows', 'dense'])
def test_data_parallel_step_over_rccl_one_rank(exchange, monkeypatch, tmp_path):
    """The data-parallel step with the real RCCL backend on the one GPU of the test box: a one-rank ``nccl`` process group and
    SBR_FORCE_DIST=1 make the engine run its multi-GPU launch structure (two graphs, asynchronous all-reduce of the user part
    on RCCL's stream between them, all-reduce of the rest, wait on the compute stream, division by the world size) and the
    item-sharded scoring its top-k all-gather + merge. Summing over one rank changes nothing, so losses, parameters and
    top-k lists must equal the run without a process group. (More ranks need more GPUs: the driver's scaling run.)
    ``sparse_user_rows``: the lookup user side all-gathers (table row, gradient row) pairs and scatters them after the
    exchange instead of all-reducing the dense table gradient; one batch is ragged (unused capacity of the send buffer)."""
    import torch.distributed as dist
    import sibrar_amd as S
    ds = S.SyntheticDataset(300, 200, 6000, item_dense={'text': 40}, seed=3, n_negative_samples=3)
    cfg = {'shared_common_dim': 64, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'item_embedding'}],
                    'single_branch_hidden_layers': [64], 'preference_hidden_layers': [], 'common_modality_dim': 64}}

    def run():
        torch.manual_seed(11)
        np.random.seed(11)
        net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV)
        net.train()
        opt = S.FusedOptimizer(net, 'adamw', lr=1e-2, weight_decay=1e-2)
        loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
        fused = S.FusedTrainStep(net, loss, opt, use_graph=True)
        rng = np.random.default_rng(9)
        losses = []
        for s_ in range(10):
            B = 64 if s_ != 5 else 40
            u = torch.from_numpy(rng.integers(0, ds.n_users, size=B))
            i = torch.from_numpy(rng.integers(0, ds.n_items, size=(B, 4)))
            labels = torch.zeros(B, 4, dtype=torch.float64)
            labels[:, 0] = 1
            losses.append(torch.stack(fused.step(u, i, labels)).cpu())
        net.eval()
        with torch.no_grad():
            i16 = S.ops.cast_f16(net.get_item_representations(torch.arange(ds.n_items, device=DEV)))
            users = torch.arange(ds.n_users, device=DEV)
            u16 = S.ops.cast_f16(net.get_user_representations(users))
            val0, idx0 = S.ops.score_topk_f16(u16, i16, 10, users, None, None, item_offset=0)
            val, idx = S.parallel.all_gather_topk(val0, idx0, 10)         # one shard: the merge must return the list itself
        torch.cuda.synchronize()
        fused.close()
        assert torch.equal(val, val0) and torch.equal(idx.to(idx0.dtype), idx0)
        return fused, losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}

    monkeypatch.setenv('SBR_FORCE_SPLIT', '1')
    monkeypatch.setenv('SBR_SPARSE_EXCHANGE', '1' if exchange == 'sparse_user_rows' else '0')
    monkeypatch.setenv('SBR_DEFERRED_ADAM', '1' if exchange == 'sparse_user_rows' else '0')   # row-wise Adam over the gathered row lists
    plain = run()
    monkeypatch.setenv('SBR_FORCE_DIST', '1')
    dist.init_process_group('nccl', init_method=f'file://{tmp_path}/rdzv', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        assert S.parallel.is_distributed()
        rccl = run()
    finally:
        dist.destroy_process_group()
    assert rccl[0].split and rccl[0].n_replays >= 4
    assert bool(rccl[0]._sparse) == (exchange == 'sparse_user_rows') and not plain[0]._sparse
    for s_, (a, b) in enumerate(zip(plain[1], rccl[1])):
        close(b, a, what=f'losses step {s_}', rtol=1e-6, atol=1e-9)
    skip = set(bn_shadowed_biases(plain[2].keys())) | {'item_embedding_module.sb_net.1.bias'}
    for k in plain[2]:
        if k not in skip:
            close(rccl[2][k].double(), plain[2][k].double(), what=k, rtol=1e-4, atol=1e-6, norm_rtol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('optimizer', ['adamw', 'adam'])
def test_fused_step_with_deferred_row_wise_adam_equals_dense_optimizer(optimizer, monkeypatch):
    """The fused step updates a lookup user table row by row (engine.DeferredTable / sbr_adam_rows: rows without gradient take
    their zero-gradient steps later, in order; bit-exactness of that replay is pinned on deterministic gradients in
    tests/test_hip_kernels.py). End to end — 25 steps in which most of 2000 users are touched rarely, one duplicated user per
    batch, a state_dict() flush in the middle — the run agrees with the dense-optimizer run (SBR_DEFERRED_ADAM=0).
    (Users are otherwise distinct within a batch: with three or more slots of one user the float atomics of the table gradient
    add in varying order, and in this tiny world that last-bit noise occasionally flips a ReLU gate or the sign of a near-zero
    Adam step — two dense runs then differ by 0.08 in a parameter in ~5 % of runs, tools/lab/deferred_check.py.)"""
    import sibrar_amd as S
    ds = S.SyntheticDataset(2000, 200, 9000, item_dense={'text': 40}, seed=3, n_negative_samples=3)
    cfg = {'shared_common_dim': 32, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'item_embedding'}],
                    'single_branch_hidden_layers': [32], 'preference_hidden_layers': [], 'common_modality_dim': 32}}
    runs = []
    for deferred in ('0', '1'):
        monkeypatch.setenv('SBR_DEFERRED_ADAM', deferred)
        torch.manual_seed(11)
        np.random.seed(11)
        net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV)
        net.train()
        opt = S.FusedOptimizer(net, optimizer, lr=1e-2, weight_decay=1e-2)
        loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
        fused = S.FusedTrainStep(net, loss, opt)
        assert (fused.deferred is not None) == (deferred == '1')
        rng = np.random.default_rng(9)
        losses, mid = [], None
        for s_ in range(25):
            u = torch.from_numpy(rng.permutation(60 if s_ % 5 == 0 else ds.n_users)[:48].copy())   # some batches hit few users
            u[1] = u[0]                                                       # a duplicate row inside the batch
            i = torch.from_numpy(rng.integers(0, ds.n_items, size=(48, 4)))
            labels = torch.zeros(48, 4, dtype=torch.float64)
            labels[:, 0] = 1
            losses.append(torch.stack(fused.step(u, i, labels)).cpu())
            if s_ == 12:
                mid = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        fused.close()
        lo, hi = fused._urange
        runs.append((losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()},
                     opt.m[lo:hi].cpu().clone(), opt.v[lo:hi].cpu().clone(), mid))
    key = 'user_embedding_module.embedding_layer.weight'
    close(runs[1][1][key], runs[0][1][key], what='user table', rtol=1e-4, atol=1e-5)
    close(runs[1][4][key], runs[0][4][key], what='user table at the mid-run flush', rtol=1e-4, atol=1e-5)
    close(runs[1][2], runs[0][2], what='first moments', rtol=1e-4, atol=1e-6)
    close(runs[1][3], runs[0][3], what='second moments', rtol=1e-4, atol=1e-8)
    for s_, (a_, b_) in enumerate(zip(runs[0][0], runs[1][0])):
        close(b_, a_, what=f'losses step {s_}', rtol=1e-5, atol=1e-8)


@pytest.mark.gpu
def test_two_rank_data_parallel_rehearsal_on_one_gpu():
    """The whole multi-process path of bench.py with TWO ranks sharing this box's one GPU (gloo carries the collectives; RCCL
    refuses two ranks on one device): rank-local sampling, two-phase step, all-gather exchange of the lookup user rows with the
    deterministic sorted scatter, dense all-reduce of the rest, item-sharded scoring with the top-k all-gather + merge kernel.
    The replicas' parameters must agree exactly after the run and training must have made progress."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_PORT')}
    env['SBR_DIST_BACKEND'] = 'gloo'
    # no external launcher: `bench.py --gpus 2` starts its two ranks itself (bench.launch_ranks)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--small', '--steps', '20', '--warmup', '3', '--no-b256']
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'dp2'
    assert out['config']['ranks_seen'] == 2 and out['config']['backend'].startswith('gloo')
    assert out['config']['replica_param_checksum_spread'] == 0.0
    assert 'all-gather' in out['config']['user_table_gradient_exchange']
    assert 0.0 < out['config']['loss_after_timed_steps'] < 2.45          # ln(11) = 2.40 at initialisation, falling
    assert out['scoring']['sharding'] == 'items/2' and out['scoring']['value'] > 0
    # the two BASELINE configs that name 8 GPUs ride in the multi-GPU line at their own shapes (reduced user / item counts under --small):
    # c4 data-parallel with the sparse row exchange, c5 item-sharded with the top-k all-gather + merge
    c4, c5 = out['c4_dp'], out['c5']
    assert c4['replica_param_checksum_spread'] == 0.0
    for b in ('b256', 'b8192'):
        assert 'all-gather' in c4[b]['user_table_gradient_exchange'] and c4[b]['value'] > 0
        assert 0.0 < c4[b]['loss_after_timed_steps'] < 2.45
    assert c5['sharding'] == 'items/2' and c5['lists_sorted_and_in_range'] and c5['value'] > 0 and c5['chunks'] == 3
    assert len(line) < 8000, f'the bench line is {len(line)} bytes: the record keeps an 8 KB tail'


@pytest.mark.gpu
def test_loader_pipeline_equals_inline_steps():
    """Batches prepared ahead by the loader's two producer threads (collate -> FusedTrainStep.prepare with pinned packed
    uploads, device-cached labels, graph replay) train the model exactly like the same batches stepped inline."""
    import sibrar_amd as S
    cfg = {'shared_common_dim': 32, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'item_embedding'}],
                    'single_branch_hidden_layers': [32], 'preference_hidden_layers': [], 'common_modality_dim': 32}}
    finals = []
    for piped in (False, True):
        ds = S.SyntheticDataset(300, 200, 6000, item_dense={'text': 40}, seed=3, n_negative_samples=3)
        torch.manual_seed(11)
        np.random.seed(11)
        net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV)
        net.train()
        opt = S.FusedOptimizer(net, 'adamw', lr=1e-2, weight_decay=1e-2)
        loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
        fused = S.FusedTrainStep(net, loss, opt, use_graph=piped)
        loader = S.NegativeSamplingDataLoader(ds, batch_size=64, shuffle=True, device=DEV, max_batches=12,
                                              prefetch=2 if piped else 0, prepare_fn=fused.prepare if piped else None)
        losses = [torch.stack(fused.step(*b)).cpu() for b in loader]
        loader.close()
        assert len(losses) == 12
        if piped:
            assert fused.n_replays >= 4 and len(fused._label_cache) == 1
        finals.append((losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
    for s_, (a, b) in enumerate(zip(finals[0][0], finals[1][0])):
        close(b, a, what=f'losses step {s_}', rtol=1e-6, atol=1e-9)
    skip = set(bn_shadowed_biases(finals[0][1].keys())) | {'item_embedding_module.sb_net.1.bias'}
    for k in finals[0][1]:
        if k not in skip:
            # AdamW (lr 1e-2, 12 steps) amplifies summation-order noise of small gradients: a few 1e-5 on single elements
            close(finals[1][1][k].double(), finals[0][1][k].double(), what=k, rtol=1e-4, atol=1e-6, norm_rtol=1e-3)


@pytest.mark.gpu
def test_trainer_fit_end_to_end(tmp_path):
    """Trainer.fit (train/trainer.py:98-170): initial validation, epochs over the loader (prepared batches, graph replay),
    validation after every epoch, best-model checkpoint, patience — on a learnable synthetic world: the loss must fall and
    NDCG@10 must rise above the untrained model's, and the saved checkpoint must reload into a fresh model."""
    import sibrar_amd as S
    ds = S.SyntheticDataset(400, 150, 9000, item_dense={'text': 24}, item_tags={'genres': (12, 3)}, seed=5, n_negative_samples=5,
                            holdout_per_user=2)
    cfg = {'shared_common_dim': 32, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'genres'}, {'feature_name': 'item_embedding'}],
                    'single_branch_hidden_layers': [32], 'preference_hidden_layers': [], 'common_modality_dim': 32,
                    'embedding_regularization_type': 'pairwise_single', 'regularization_temperature': 0.5,
                    'regularization_weight': 1e-2}}
    torch.manual_seed(3)
    np.random.seed(3)
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV)
    loss = S.RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=5)
    ev = ds.eval_view()
    conf = {'learn': {'lr': 5e-3, 'wd': 1e-6, 'optimizer': 'adamw', 'n_epochs': 4, 'optimizing_metric': 'ndcg@10', 'max_patience': 3},
            'run_settings': {'device': DEV, 'batch_verbose': False}, 'results_path': str(tmp_path),
            'eval': S.evaluation._Cfg(top_k=(1, 10, 20)), 'train_eval': None, 'scorer': 'fp16_fused' if False else 'fp32'}
    train_loader = S.NegativeSamplingDataLoader(ds, batch_size=256, shuffle=True, device=DEV)
    val_loader = type('L', (), {'dataset': ev, 'batch_size': 128})()
    tr = S.Trainer(net, train_loader, val_loader, loss, conf)
    assert tr.fused is not None and train_loader.prepare_fn is not None and train_loader.prefetch > 0
    first = tr.val()['ndcg@10']
    l0 = tr.train()['train/loss']
    best = tr.fit()
    train_loader.close()
    l_end = tr.train()['train/loss']
    train_loader.close()
    assert l_end < l0, (l0, l_end)
    assert best['ndcg@10'] >= first and best['max_optimizing_metric'] == best['ndcg@10'] and 'best_epoch' in best
    assert tr.fused.n_replays > 0
    fresh = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV)
    fresh.load_model_from_path(str(tmp_path))
    sd_a, sd_b = fresh.state_dict(), None
    assert set(sd_a) == set(net.state_dict())


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['mf', 'ifeatmf', 'ufeatmf', 'dropoutnet', 'sgdbias'])
def test_sibling_models_train_and_evaluate_through_trainer(model, tmp_path):
    """The sibling models of SURVEY 8(f).4 through the same Trainer / loader / evaluation as SingleBranchNet (autograd over the
    HIP kernels, fused optimizer, full-catalogue evaluation via their own combine): the loss falls, NDCG@10 does not get worse
    than the untrained model's, the best checkpoint reloads."""
    import sibrar_amd as S
    ds = S.SyntheticDataset(400, 150, 9000, item_dense={'text': 24}, item_tags={'genres': (12, 3)}, seed=5, n_negative_samples=5,
                            holdout_per_user=2)
    torch.manual_seed(3)
    np.random.seed(3)
    common = dict(aggregate_for_rec=True, lambda_content=1e-4, temperature=0.5, embedding_loss_aggregator='mean',
                  intermediate_layers=[16], embedding_dim=16, use_user_bias=False, use_item_bias=True, use_global_bias=False)
    confs = {
        'mf': dict(embedding_dim=16, use_user_bias=False, use_item_bias=True, use_global_bias=True),
        'ifeatmf': dict(feature_name='text', **common),
        'ufeatmf': dict(feature_name='user_embedding', **{**common, 'intermediate_layers': None}),
        'dropoutnet': dict(user=dict(features=[], preference_layers=[32], common_hidden_layers=[]),
                           item=dict(features=[dict(feature_name='text', embedding_dim=16), dict(feature_name='genres', embedding_dim=8)],
                                     preference_layers=[32], common_hidden_layers=[32]), shared_common_dim=16),
        'sgdbias': {},
    }
    if model == 'ufeatmf':        # a categorical user feature: the user id itself (what SingleBranchNet adds as 'user_embedding')
        ds.user_features = dict(getattr(ds, 'user_features', {}) or {})
        ds.user_features['user_embedding'] = S.HostFeature('user_embedding', 'categorical', np.arange(ds.n_users), n_categories=ds.n_users)
    net = S.ALGORITHMS[model].build_from_conf(confs[model], ds).to(DEV)
    loss = S.RecBinaryCrossEntropy(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=5)
    # DropoutNet ends in a ReLU on both sides (PolyLinear's default output_fn): at lr 1e-2 the output units die within an epoch
    # (all logits 0, loss ln 2 from then on — the reference's arithmetic does the same), so it trains at 1e-3
    lr = 1e-3 if model == 'dropoutnet' else 1e-2
    conf = {'learn': {'lr': lr, 'wd': 1e-6, 'optimizer': 'adamw', 'n_epochs': 3, 'optimizing_metric': 'ndcg@10', 'max_patience': 3},
            'run_settings': {'device': DEV, 'batch_verbose': False}, 'results_path': str(tmp_path),
            'eval': S.evaluation._Cfg(top_k=(1, 10, 20)), 'train_eval': None, 'scorer': 'fp32'}
    train_loader = S.NegativeSamplingDataLoader(ds, batch_size=256, shuffle=True, device=DEV)
    val_loader = type('L', (), {'dataset': ds.eval_view(), 'batch_size': 128})()
    tr = S.Trainer(net, train_loader, val_loader, loss, conf)
    assert tr.fused is None
    l0 = tr.train()['train/loss']
    best = tr.fit()
    l_end = tr.train()['train/loss']
    train_loader.close()
    assert l_end < l0, (l0, l_end)
    assert best['ndcg@10'] > 0 and best['max_optimizing_metric'] == best['ndcg@10'] and 'best_epoch' in best
    fresh = S.ALGORITHMS[model].build_from_conf(confs[model], ds).to(DEV)
    fresh.load_model_from_path(str(tmp_path))
    assert set(fresh.state_dict()) == set(net.state_dict())


@pytest.mark.gpu
def test_split_dataset_trains_and_evaluates():
    """The on-disk fixture (tests/golden/split_random, the reference's directory format) loaded with load_split_dataset drives the
    whole path: SingleBranchNet over a tag + vector + id-embedding item entity and a categorical user feature, loader,
    fused steps, full evaluation of the val split with the train interactions excluded."""
    import os
    import sibrar_amd as S
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'split_random')
    fdefs = dict(user_feature_definitions=[{'name': 'gender', 'type': 'categorical'}],
                 item_feature_definitions=[{'name': 'genres', 'type': 'tag', 'tag_split_sep': '|'}, {'name': 'text', 'type': 'vector'}])
    train = S.load_split_dataset(here, 'train', n_negative_samples=3, **fdefs)
    val = S.load_split_dataset(here, 'val', **fdefs)
    cfg = {'shared_common_dim': 16, 'user': {'feature_name': 'gender', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'genres'}, {'feature_name': 'text'}, {'feature_name': 'item_embedding'}],
                    'single_branch_hidden_layers': [16], 'preference_hidden_layers': [], 'common_modality_dim': 16}}
    torch.manual_seed(1)
    np.random.seed(1)
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), train).to(DEV)
    loss = S.RecBayesianPersonalizedRankingLoss(n_items=train.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
    opt = S.FusedOptimizer(net, 'adamw', lr=1e-2, weight_decay=0.)
    fused = S.FusedTrainStep(net, loss, opt)
    net.train()
    loader = S.NegativeSamplingDataLoader(train, batch_size=16, shuffle=True, device=DEV)
    vals = [float(fused.step(*b)[0]) for _ in range(3) for b in loader]
    assert len(vals) == 3 * len(loader) and all(np.isfinite(vals))
    ev = S.FullEvaluator(config=S.evaluation._Cfg(top_k=(1, 5)), dataset=val)
    res = S.evaluate_recommender_algorithm(net, type('L', (), {'dataset': val, 'batch_size': 8})(), ev, DEV)
    assert 0.0 <= res['ndcg@5'] <= 1.0 and 'recall@1' in res


@pytest.mark.gpu
def test_missing_feature_row_raises_keyerror_not_a_fault():
    """An item id without a row in one of its features (the reference: KeyError from Feature.__getitem__'s dict lookup,
    data/Feature.py:146). The kernels must stay in bounds (row 0 is substituted) and the host must raise KeyError at its next
    check — also for ids beyond the end of the id -> row map."""
    import sibrar_amd as S
    ds = S.SyntheticDataset(50, 30, 400, item_dense={'text': 8}, seed=2, n_negative_samples=3)
    f = ds.item_features['text']
    keep = np.arange(0, 20)                                   # items 20..29 have no 'text' row
    ds.item_features['text'] = S.features.HostFeature('text', 'dense', np.asarray(f.values)[keep], indices=keep)
    cfg = {'shared_common_dim': 8, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'item_embedding'}],
                    'single_branch_hidden_layers': [8], 'preference_hidden_layers': [], 'common_modality_dim': 8}}
    torch.manual_seed(0)
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV)
    net.train()
    loss = S.RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
    fused = S.FusedTrainStep(net, loss, S.FusedOptimizer(net, 'adamw', lr=1e-3, weight_decay=0.))
    u = torch.arange(16)
    ok_items = torch.from_numpy(np.random.default_rng(0).integers(0, 20, size=(16, 4)))
    labels = torch.zeros(16, 4, dtype=torch.float64)
    for _ in range(3):
        fused.step(u, ok_items, labels)
    fused.check_errors()                                      # all ids covered: nothing raised
    bad_items = ok_items.clone()
    bad_items[3, 2] = 27                                      # beyond the text map (length 20)
    for _ in range(3):                                        # enough steps for 'text' to be drawn for that slot
        fused.step(u, bad_items, labels)
    out = fused.step(u, bad_items, labels)
    assert torch.isfinite(torch.stack(list(out))).all()
    with pytest.raises(KeyError):
        for _ in range(20):
            fused.step(u, bad_items, labels)
        fused.check_errors()
    fused.check_errors()                                      # the flag was reset by the raising check
    net.eval()
    with pytest.raises(KeyError):
        net.get_item_representations(torch.arange(30, device=DEV))
        net.check_index_errors()


@pytest.mark.gpu
def test_fused_step_with_input_dropout_replays_a_graph():
    """single_branch_input_dropout > 0 (sgd_alg.py:1815; most shipped sbnet configs use 0.02 or 0.2): the step's dropout seed
    lives in device memory (part of the batch upload), so the step is still captured and replayed; graph replay and plain
    launches must produce the same trajectory (same seeds -> same masks), and eval mode must be deterministic."""
    import sibrar_amd as S
    ds = S.SyntheticDataset(200, 120, 4000, item_dense={'text': 24}, seed=4, n_negative_samples=3)
    cfg = {'shared_common_dim': 16, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'item_embedding'}], 'single_branch_hidden_layers': [16],
                    'preference_hidden_layers': [], 'common_modality_dim': 16, 'single_branch_input_dropout': 0.3,
                    'normalize_single_branch_input': True}}
    rng = np.random.default_rng(0)
    u = torch.from_numpy(rng.integers(0, ds.n_users, size=64))
    i = torch.from_numpy(rng.integers(0, ds.n_items, size=(64, 4)))
    labels = torch.zeros(64, 4, dtype=torch.float64)
    runs = []
    for use_graph in (False, True):
        torch.manual_seed(5)
        np.random.seed(5)
        net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV)
        net.train()
        loss = S.RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
        fused = S.FusedTrainStep(net, loss, S.FusedOptimizer(net, 'adamw', lr=1e-2, weight_decay=0.), use_graph=use_graph)
        vals = [float(fused.step(u, i, labels)[0]) for _ in range(12)]
        assert all(np.isfinite(vals)) and vals[-1] < vals[0]
        assert (fused.n_replays > 0) == use_graph
        runs.append(vals)
    assert len(set(round(v, 6) for v in runs[0])) > 6                    # the masks change from step to step
    close(torch.tensor(runs[1]), torch.tensor(runs[0]), what='dropout trajectories graph vs plain', rtol=1e-4, atol=1e-6)
    net.eval()
    a = net.get_item_representations(torch.arange(ds.n_items, device=DEV))
    b = net.get_item_representations(torch.arange(ds.n_items, device=DEV))
    assert torch.equal(a, b)


_VARIANTS = {
    'central_max_normalize': dict(embedding_regularization_type='central_modality', central_modality='text', aggregation_fn='max',
                                  normalize_single_branch_input=True, regularization_temperature=0.3, regularization_weight=0.05),
    'noreg_bn_every2_tanh_outact': dict(apply_batch_norm_every=2, apply_output_activation=True, activation_fn='tanh',
                                        single_branch_hidden_layers=[24, 16, 16]),
    'pairwise_nobn_sigmoid': dict(embedding_regularization_type='pairwise_single', apply_batch_normalization=False,
                                  activation_fn='sigmoid', regularization_weight=0.1),
    'noreg_bn_last_selu': dict(apply_batch_norm_every=-1, activation_fn='selu', single_branch_hidden_layers=[16, 16]),
}


@pytest.mark.gpu
@pytest.mark.parametrize('variant', sorted(_VARIANTS))
@pytest.mark.parametrize('user_entity', [False, True], ids=['user_lookup', 'user_entity'])
def test_fused_step_matches_module_path_on_config_variants(variant, user_entity):
    """engine.FusedTrainStep == the nn.Module / autograd path (pinned by the golden groups) over the entity options the shipped
    configs use: regularisation types, max aggregation, input normalisation, BatchNorm placement (trailing / every n / last / off),
    output activation, all four activations, tag + dense + CSR-interactions + id modalities, entity or lookup user side."""
    import sibrar_amd as S
    ds = S.SyntheticDataset(120, 80, 2500, item_dense={'text': 12}, item_tags={'genres': (9, 3)}, user_categorical={'gender': 3},
                            seed=6, n_negative_samples=3)
    item = {'features': [{'feature_name': 'text', 'feature_hidden_layers': [10]}, {'feature_name': 'genres'},
                         {'feature_name': 'interactions'}, {'feature_name': 'item_embedding'}],
            'single_branch_hidden_layers': [16], 'preference_hidden_layers': [], 'common_modality_dim': 16}
    item.update(_VARIANTS[variant])
    user = ({'features': [{'feature_name': 'interactions'}, {'feature_name': 'gender'}], 'single_branch_hidden_layers': [],
             'preference_hidden_layers': [], 'common_modality_dim': 16, 'embedding_regularization_type': 'pairwise_single',
             'regularization_weight': 0.02} if user_entity else {'feature_name': 'user_embedding', 'embedding_dim': -1})
    cfg = {'shared_common_dim': 16, 'user': user, 'item': item}
    nets = []
    for _ in range(2):
        torch.manual_seed(21)
        np.random.seed(21)
        nets.append(S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV).train())
    opts = [S.FusedOptimizer(n_, 'adamw', lr=1e-2, weight_decay=1e-3) for n_ in nets]
    lossf = S.RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
    fused = S.FusedTrainStep(nets[1], lossf, opts[1])
    rng = np.random.default_rng(8)
    for s_ in range(4):
        u = torch.from_numpy(rng.integers(0, ds.n_users, size=40))
        i = torch.from_numpy(rng.integers(0, ds.n_items, size=(40, 4)))
        labels = torch.zeros(40, 4, dtype=torch.float64)
        labels[:, 0] = 1
        logits = nets[0](u.to(DEV), i.to(DEV))
        loss = lossf.compute_loss(logits, labels.to(DEV))
        reg = nets[0].get_and_reset_other_loss()['reg_loss']
        (loss + reg.to(loss.device).sum()).backward()
        opts[0].step()
        opts[0].zero_grad()
        total, rec, reg2 = fused.step(u, i, labels)
        close(rec.cpu(), loss.detach().cpu().double(), what=f'rec loss step {s_}', rtol=2e-5, atol=1e-7)
        close(reg2.cpu().reshape(-1), reg.detach().cpu().double().reshape(-1), what=f'reg loss step {s_}', rtol=2e-5, atol=1e-7)
    nets[1].check_index_errors()
    sd0, sd1 = nets[0].state_dict(), nets[1].state_dict()
    # zero-gradient parameters (Adam turns rounding noise into +-lr steps): biases in front of a BatchNorm, and the shift of a
    # BatchNorm that ends the item network — BPR differences cancel a common shift of all item representations
    skip = set(bn_shadowed_biases(sd0.keys())) | {'item_embedding_module.sb_net.1.bias',
                                                  'item_embedding_module.sb_net.0.layers.batch_norm.bias'}
    for k in sd0:
        if k not in skip:
            close(sd1[k].cpu(), sd0[k].cpu(), what=k, rtol=1e-4, atol=1e-6, norm_rtol=1e-3)


@pytest.mark.gpu
def test_workspace_regrowth_after_capture_does_not_corrupt_replays():
    """The split-K / InfoNCE scratch buffers are shared by every launch of a process and grow on demand, but their addresses
    are baked into captured step graphs. A later, larger product (another model, an autograd-path call, a new padded
    signature) outgrows the buffer: the old block must stay allocated — a replay that wrote its slabs into memory the caching
    allocator had handed to somebody else would corrupt that tensor (and read back garbage itself). Capture small, outgrow
    the workspace, put a canary where the allocator would have recycled the block, replay: trajectory == plain launches and
    the canary is intact."""
    import sibrar_amd as S
    ds = S.SyntheticDataset(300, 200, 6000, item_dense={'text': 40}, seed=3, n_negative_samples=3)
    cfg = {'shared_common_dim': 32, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'item_embedding'}],
                    'single_branch_hidden_layers': [32], 'preference_hidden_layers': [], 'common_modality_dim': 32}}
    rng0 = np.random.default_rng(9)
    batches = []
    for _ in range(10):
        labels = torch.zeros(64, 4, dtype=torch.float64)
        labels[:, 0] = 1
        batches.append((torch.from_numpy(rng0.integers(0, ds.n_users, size=64)), torch.from_numpy(rng0.integers(0, ds.n_items, size=(64, 4))),
                        labels))
    runs = []
    for use_graph in (False, True):
        torch.manual_seed(11)
        np.random.seed(11)
        net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV)
        net.train()
        opt = S.FusedOptimizer(net, 'adamw', lr=1e-2, weight_decay=1e-2)
        loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
        fused = S.FusedTrainStep(net, loss, opt, use_graph=use_graph)
        losses = []
        canary = None
        for s_, b in enumerate(batches):
            if s_ == 6 and use_graph:
                assert fused.n_replays > 0
                dev = torch.device(DEV, torch.cuda.current_device())
                old = S.ops._TN_WS[next(iter(S.ops._TN_WS))]
                gen = S.ops.WS_GENERATION
                # a dW product whose slabs need far more than the current workspace: [R, 512]^T [R, 512] over 200k rows
                R = 200_000
                big = S.ops.matmul_tn(torch.ones(R, 512, device=DEV), torch.ones(R, 512, device=DEV))
                assert S.ops.WS_GENERATION > gen, 'the product did not outgrow the workspace: enlarge R'
                assert float(big[0, 0]) == R
                del big
                torch.cuda.synchronize()
                # whatever the allocator hands out next must not be the retired block
                canary = [torch.full((old.numel(),), 7.0, device=DEV) for _ in range(3)]
                assert all(c.data_ptr() != old.data_ptr() for c in canary)
            losses.append(torch.stack(fused.step(*b)).cpu())
        if canary is not None:
            torch.cuda.synchronize()
            assert all(bool((c == 7.0).all()) for c in canary)
        runs.append((losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
        fused.close()
    for s_, (a, b) in enumerate(zip(runs[0][0], runs[1][0])):
        close(b, a, what=f'losses step {s_}', rtol=1e-6, atol=1e-9)
    skip = set(bn_shadowed_biases(runs[0][1].keys())) | {'item_embedding_module.sb_net.1.bias'}
    for k in runs[0][1]:
        if k not in skip:
            close(runs[1][1][k].double(), runs[0][1][k].double(), what=k, rtol=1e-4, atol=1e-6, norm_rtol=1e-3)
