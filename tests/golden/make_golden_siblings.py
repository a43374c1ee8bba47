#!/usr/bin/env python3
"""G14: sibling models that reuse the hot path's kernels, generated with the REAL reference.

    PYTHONHASHSEED=0 python tests/golden/make_golden_siblings.py      (build container only)

SGDMatrixFactorization (algorithms/sgd_alg.py:126-200) with every bias combination used by the reference's configs, and
SGDBaseline (:88-123): parameters, one batch, train-mode logits, BPR loss (train/rec_losses.py:63-83), the gradient of every
parameter, and eval-mode all-pairs scores through get_*_representations + combine (the evaluation path, eval/eval.py:205-217).
Only data is written.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the import placeholders, asserts PYTHONHASHSEED=0)

import torch  # noqa: E402
from algorithms.sgd_alg import (SGDMatrixFactorization, SGDBaseline, ItemFeatureMatrixFactorization,  # noqa: E402
                                UserFeatureMatrixFactorization)
from train.rec_losses import RecBayesianPersonalizedRankingLoss  # noqa: E402

U, I = G.U, G.I
arrays, meta = {}, {'cases': []}
u, i, labels = G.batch(14)
arrays['u'], arrays['i'], arrays['labels'] = G.t2n(u), G.t2n(i), G.t2n(labels)
loss_fn = RecBayesianPersonalizedRankingLoss(n_items=I, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)


def record(name, m):
    with torch.no_grad():
        for p_name, p in m.named_parameters():
            if 'bias' in p_name:                      # biases initialise to ~0: give them signal
                p.copy_(torch.randn_like(p) * 0.3)
    arrays.update(G.sd2n(m.state_dict(), f'{name}/sd/'))
    m.train()
    logits = m(u, i)
    loss = loss_fn.compute_loss(logits, labels)
    loss.backward()
    arrays[f'{name}/logits'] = G.t2n(logits)
    arrays[f'{name}/loss'] = G.t2n(loss)
    for p_name, p in m.named_parameters():
        arrays[f'{name}/grad/{p_name}'] = G.t2n(p.grad)
    m.eval()
    with torch.no_grad():
        all_items = torch.arange(I)
        ir = m.get_item_representations(all_items)
        ur = m.get_user_representations(u)
        if name.startswith('mf'):
            # eval/eval.py:205-217 path: item representations once, combine per user batch
            i_arg = tuple(r[None] if r.ndim < 2 else r[None] for r in ir) if isinstance(ir, tuple) else ir[None]
            arrays[f'{name}/scores_all'] = G.t2n(m.combine_user_item_representations(ur, i_arg))
        else:
            arrays[f'{name}/scores_all'] = G.t2n(m(u, all_items[None].expand(len(u), -1)))


# use_user_bias=True raises in the reference's combine (sgd_alg.py:190: a [B, 1] bias indexed [:, None] is added in place to [B, N])
for ub, ib, gb in ((False, False, False), (False, True, True), (False, True, False), (False, False, True)):
    torch.manual_seed(14)
    name = f'mf_u{int(ub)}i{int(ib)}g{int(gb)}'
    record(name, SGDMatrixFactorization(U, I, embedding_dim=8, use_user_bias=ub, use_item_bias=ib, use_global_bias=gb))
    meta['cases'].append({'name': name, 'use_user_bias': ub, 'use_item_bias': ib, 'use_global_bias': gb, 'embedding_dim': 8})
torch.manual_seed(15)
record('baseline', SGDBaseline(U, I))

# hybrid factorisation models (sgd_alg.py:1399-1614) on the shared world: item text (dense 768-like vector), item genres (tags),
# user gender (categorical)
arrays.update(G.world_arrays())
ds = G.make_dataset()


def record_hybrid(name, m):
    arrays.update(G.sd2n(m.state_dict(), f'{name}/sd/'))
    m.train()
    logits = m(u, i)
    reg = m.get_and_reset_other_loss()['reg_loss']
    loss = loss_fn.compute_loss(logits, labels) + reg
    loss.backward()
    arrays[f'{name}/logits'], arrays[f'{name}/loss'] = G.t2n(logits), G.t2n(loss)
    arrays[f'{name}/reg_loss'] = G.t2n(torch.as_tensor(reg, dtype=torch.float32).reshape(-1))
    for p_name, p in m.named_parameters():
        arrays[f'{name}/grad/{p_name}'] = G.t2n(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)
    m.eval()
    with torch.no_grad():
        ir = m.get_item_representations(torch.arange(I))
        arrays[f'{name}/scores_all'] = G.t2n(m.combine_user_item_representations(m.get_user_representations(u), ir))


hybrids = [
    ('ifmf_text', ItemFeatureMatrixFactorization, dict(feature_name='text', aggregate_for_rec=False, temperature=0.5,
                                                      intermediate_layers=[10], embedding_dim=8)),
    ('ifmf_text_agg_bias', ItemFeatureMatrixFactorization, dict(feature_name='text', aggregate_for_rec=True, temperature=0.1,
                                                               embedding_loss_aggregator='sum', intermediate_layers=None,
                                                               embedding_dim=8, use_item_bias=True, use_global_bias=True)),
    ('ifmf_genres', ItemFeatureMatrixFactorization, dict(feature_name='genres', aggregate_for_rec=True, temperature=1.0,
                                                        intermediate_layers=None, embedding_dim=8)),
    ('ufmf_gender', UserFeatureMatrixFactorization, dict(feature_name='gender', aggregate_for_rec=True, temperature=0.5,
                                                        intermediate_layers=None, embedding_dim=8)),
    ('ufmf_gender_plain', UserFeatureMatrixFactorization, dict(feature_name='gender', aggregate_for_rec=False, temperature=0.5,
                                                              intermediate_layers=None, embedding_dim=8, use_item_bias=True)),
]
for name, cls, kw in hybrids:
    torch.manual_seed(16)
    record_hybrid(name, cls(ds, **kw))
    meta['cases'].append({'name': name, 'class': cls.__name__, 'kwargs': kw})
np.savez_compressed(os.path.join(HERE, 'g14_sibling_models.npz'), **arrays)
json.dump(meta, open(os.path.join(HERE, 'g14_sibling_models.json'), 'w'), indent=1)
print(len(arrays), [c['name'] for c in meta['cases']])


# ---- DropoutNet (sgd_alg.py:1617-1762) on the shared world ----------------------------------------------------------------------
def g15_dropoutnet():
    from algorithms.sgd_alg import DropoutNet
    from data.dataset import InteractionRecDataset
    from data.module_config_classes import DropoutNetConfig, DropoutNetEntityConfig, FeatureModuleConfig
    out, cases = dict(G.world_arrays()), []
    ds2 = G.make_dataset()
    ds2._get_numpy_array = InteractionRecDataset._get_numpy_array
    ds2.get_user_interaction_vectors = lambda idx: InteractionRecDataset._get_interaction_vectors(ds2, 'user', idx)
    ds2.get_item_interaction_vectors = lambda idx: InteractionRecDataset._get_interaction_vectors(ds2, 'item', idx)
    out['u'], out['i'], out['labels'] = G.t2n(u), G.t2n(i), G.t2n(labels)
    confs = {
        'dn_small': dict(user=dict(features=[dict(feature_name='gender', embedding_dim=4)], preference_layers=[12, 6],
                                   common_hidden_layers=[10], activation_fn='relu'),
                         item=dict(features=[dict(feature_name='text', embedding_dim=5, pre_embedding_layers=[7]),
                                             dict(feature_name='genres', embedding_dim=4)],
                                   preference_layers=[9], common_hidden_layers=[], activation_fn='tanh'),
                         shared_common_dim=8, sampling_seed=42),
        'dn_nofeat': dict(user=dict(features=[], preference_layers=[6], common_hidden_layers=[], activation_fn='relu'),
                          item=dict(features=[dict(feature_name='audio', embedding_dim=6)], preference_layers=[5, 5],
                                    common_hidden_layers=[9, 7], activation_fn='relu'),
                          shared_common_dim=6, sampling_seed=7),
    }
    for name, c in confs.items():
        def ent(d):
            return DropoutNetEntityConfig(features=[FeatureModuleConfig(**f) for f in d['features']],
                                          preference_layers=d['preference_layers'], common_hidden_layers=d['common_hidden_layers'],
                                          activation_fn=d['activation_fn'])
        cfg = DropoutNetConfig(user=ent(c['user']), item=ent(c['item']), shared_common_dim=c['shared_common_dim'],
                               sampling_seed=c['sampling_seed'])
        if not hasattr(cfg, 'to_dict'):
            type(cfg).to_dict = lambda self: {}
        torch.manual_seed(17)
        m = DropoutNet(cfg, ds2)
        out.update(G.sd2n(m.state_dict(), f'{name}/sd/'))
        # the strategies the model is about to draw (user first, then one per row of the item index matrix)
        probe = np.random.default_rng(c['sampling_seed'])
        out[f'{name}/user_strategy'] = probe.choice([1, 2], size=len(u), replace=True)
        out[f'{name}/item_strategy'] = probe.choice([1, 2], size=len(i), replace=True)
        m.train()
        logits = m(u, i)
        loss = loss_fn.compute_loss(logits, labels)
        loss.backward()
        out[f'{name}/logits'], out[f'{name}/loss'] = G.t2n(logits), G.t2n(loss)
        for p_name, p in m.named_parameters():
            out[f'{name}/grad/{p_name}'] = G.t2n(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)
        m.eval()
        with torch.no_grad():
            ir = m.get_item_representations(torch.arange(I))
            out[f'{name}/scores_all'] = G.t2n(m.combine_user_item_representations(m.get_user_representations(u), ir))
        cases.append({'name': name, 'config': c})
    np.savez_compressed(os.path.join(HERE, 'g15_dropoutnet.npz'), **out)
    json.dump({'cases': cases}, open(os.path.join(HERE, 'g15_dropoutnet.json'), 'w'), indent=1)
    print('g15', len(out), [c['name'] for c in cases], {k: out[k].tolist() for k in out if k.endswith('strategy')})


g15_dropoutnet()
