#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (the reference lives at /root/reference and never travels):

    PYTHONHASHSEED=0 python tests/golden/make_golden.py

The reference's hot-path modules are imported from where they lie (see ``_ref_import.py`` for the
import-time placeholders of packages that are not installed here); tiny synthetic inputs are pushed
through the reference's own classes and the inputs, parameters, recorded sampling decisions and
outputs are written as ``*.npz`` (+ ``manifest.json``).  Only data is written — no reference source.

Determinism: PYTHONHASHSEED=0 (the reference iterates ``list(set_of_names)``), one torch thread,
dropout disabled in every case.
"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()

import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(1)

from modules.polylinear import PolyLinear  # noqa: E402
from train.regularization_losses import InfoNCE  # noqa: E402
from utilities.utils import row_wise_sample, reproducible  # noqa: E402
from eval import metrics as ref_metrics  # noqa: E402
from data import sampling as ref_sampling  # noqa: E402
from data.Feature import Feature  # noqa: E402
from data.config_classes import FeatureDefinition, FeatureType  # noqa: E402
from data.module_config_classes import (SingleBranchNetConfig, SingleBranchNetEntityConfig,  # noqa: E402
                                        SingleBranchFeatureConfig, FeatureModuleConfig,
                                        EmbeddingRegularizationType)
from algorithms.sgd_alg import FeatureEmbedding, SingleBranchNetEntity, SingleBranchNet, SGDBaseline  # noqa: E402
from train.rec_losses import (RecBinaryCrossEntropy, RecBayesianPersonalizedRankingLoss,  # noqa: E402
                              RecSampledSoftmaxLoss)
from data.dataloader import NegativeSamplingDataLoader, TrainDataLoader, NegativeSampler  # noqa: E402

assert os.environ.get('PYTHONHASHSEED') == '0', 'run with PYTHONHASHSEED=0'

U, I = 50, 40
MANIFEST = {}


def t2n(t):
    return t.detach().cpu().numpy().copy()


def sd2n(sd, prefix='sd/'):
    return {prefix + k: t2n(v) for k, v in sd.items()}


def save(name, arrays, meta):
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **arrays)
    MANIFEST[name] = meta


# ------------------------------------------------------------------------------------------------
# synthetic dataset shared by the model cases
# ------------------------------------------------------------------------------------------------
def make_world(seed=0):
    rng = np.random.default_rng(seed)
    dense = (rng.random((U, I)) < 0.15)
    dense[:, 0] |= ~dense.any(axis=1)
    dense[0] |= ~dense.any(axis=0)
    inter = sp.csr_matrix(dense.astype(np.int8))
    text = rng.standard_normal((I, 16)).astype(np.float32)
    audio = rng.standard_normal((I, 24)).astype(np.float64)        # float64 on disk -> .float() in the reference
    tag_names = ['t%02d' % i for i in range(7)]
    genres = []
    for _ in range(I):
        n = rng.integers(1, 4)
        genres.append('|'.join(rng.choice(tag_names, size=n, replace=False)))
    gender = rng.choice(['f', 'm', 'x'], size=U).tolist()
    age = rng.choice(['a', 'b', 'c', 'd', 'e'], size=U).tolist()
    return SimpleNamespace(inter=inter, text=text, audio=audio, genres=genres, gender=gender, age=age)


W = make_world()


def make_dataset():
    item_features = {
        'text': Feature(FeatureDefinition('text', FeatureType.VECTOR), W.text),
        'audio': Feature(FeatureDefinition('audio', FeatureType.VECTOR), W.audio),
        'genres': Feature(FeatureDefinition('genres', FeatureType.TAG, tag_split_sep='|'), W.genres),
    }
    user_features = {
        'gender': Feature(FeatureDefinition('gender', FeatureType.CATEGORICAL), W.gender),
        'age': Feature(FeatureDefinition('age', FeatureType.CATEGORICAL), W.age),
    }
    return SimpleNamespace(n_users=U, n_items=I, user_features=user_features, item_features=item_features,
                           user_sampling_matrix_train=W.inter, item_sampling_matrix_train=sp.csr_matrix(W.inter.T),
                           is_cold_start_user=False, is_cold_start_item=False)


def world_arrays():
    ds = make_dataset()
    return {
        'world/inter_indptr': W.inter.indptr.astype(np.int64), 'world/inter_indices': W.inter.indices.astype(np.int64),
        'world/text': W.text, 'world/audio': W.audio,
        'world/genres_padded': np.asarray(ds.item_features['genres'].values).astype(np.int64),
        'world/genres_ntags': np.array(ds.item_features['genres'].dim),
        'world/gender': np.asarray(ds.user_features['gender'].values).astype(np.int64),
        'world/gender_ncat': np.array(ds.user_features['gender'].n_unique_categories),
        'world/age': np.asarray(ds.user_features['age'].values).astype(np.int64),
        'world/age_ncat': np.array(ds.user_features['age'].n_unique_categories),
    }


# ------------------------------------------------------------------------------------------------
# G1 PolyLinear
# ------------------------------------------------------------------------------------------------
def g1_polylinear():
    arrays, meta = {}, {'cases': []}
    cfg = [12, 10, 9, 7]
    for act in ['relu', 'tanh', 'sigmoid', 'selu']:
        for every in [0, 1, 2, -1]:
            for out_act in ([None, act] if act == 'relu' else [act]):
                name = f'{act}_bn{every}_out{out_act}'
                torch.manual_seed(hash((act, every)) % 1000)
                pl = PolyLinear(cfg, activation_fn=act, output_fn=out_act, apply_batch_norm_every=every)
                with torch.no_grad():
                    for k, v in pl.state_dict().items():
                        if k.endswith('batch_norm.weight') or ('batch_norm' in k and k.endswith('.weight')):
                            v.uniform_(0.5, 1.5)
                        if 'batch_norm' in k and k.endswith('.bias'):
                            v.uniform_(-0.5, 0.5)
                arrays.update(sd2n(pl.state_dict(), f'{name}/sd0/'))
                pl.train()
                for s in range(3):
                    x = torch.randn(11, cfg[0], requires_grad=True)
                    r = torch.randn(11, cfg[-1])
                    y = pl(x)
                    pl.zero_grad()
                    (y * r).sum().backward()
                    arrays[f'{name}/x{s}'] = t2n(x)
                    arrays[f'{name}/r{s}'] = t2n(r)
                    arrays[f'{name}/y{s}'] = t2n(y)
                    arrays[f'{name}/gx{s}'] = t2n(x.grad)
                    for k, p in pl.named_parameters():
                        arrays[f'{name}/g{s}/{k}'] = t2n(p.grad)
                arrays.update(sd2n(pl.state_dict(), f'{name}/sd3/'))
                pl.eval()
                xe = torch.randn(5, cfg[0])
                arrays[f'{name}/xe'] = t2n(xe)
                arrays[f'{name}/ye'] = t2n(pl(xe))
                meta['cases'].append({'name': name, 'layer_config': cfg, 'act': act, 'out_act': out_act,
                                      'bn_every': every})
    save('g1_polylinear', arrays, meta)


# ------------------------------------------------------------------------------------------------
# G2 FeatureEmbedding
# ------------------------------------------------------------------------------------------------
def g2_feature_embedding():
    ds = make_dataset()
    arrays, meta = world_arrays(), {'cases': []}
    inter_item = Feature(FeatureDefinition('interactions', FeatureType.VECTOR), ds.item_sampling_matrix_train)
    cases = [
        ('dense_hidden', ds.item_features['text'], dict(embedding_dim=8, pre_embedding_layers=[10]), 'text'),
        ('dense_f64', ds.item_features['audio'], dict(embedding_dim=8, pre_embedding_layers=None), 'audio'),
        ('tag', ds.item_features['genres'], dict(embedding_dim=8), 'genres'),
        ('categorical', ds.user_features['gender'], dict(embedding_dim=8), 'gender'),
        ('csr', inter_item, dict(embedding_dim=8, pre_embedding_layers=None), 'item_interactions'),
    ]
    for name, feat, kw, src in cases:
        for act in ['relu', 'tanh']:
            cname = f'{name}_{act}'
            torch.manual_seed(7)
            fe = FeatureEmbedding(feat, activation_fn=act, **kw)
            n = U if src == 'gender' else I
            idx = torch.from_numpy(np.random.default_rng(3).integers(0, n, size=(6, 4)))
            y = fe(idx)
            r = torch.randn_like(y)
            (y * r).sum().backward()
            arrays.update(sd2n(fe.state_dict(), f'{cname}/sd/'))
            arrays[f'{cname}/idx'] = t2n(idx)
            arrays[f'{cname}/y'] = t2n(y)
            arrays[f'{cname}/r'] = t2n(r)
            for k, p in fe.named_parameters():
                arrays[f'{cname}/g/{k}'] = t2n(p.grad)
            meta['cases'].append({'name': cname, 'source': src, 'act': act, 'embedding_dim': 8,
                                  'hidden': kw.get('pre_embedding_layers')})
    save('g2_feature_embedding', arrays, meta)


# ------------------------------------------------------------------------------------------------
# configs
# ------------------------------------------------------------------------------------------------
def entity_cfg(features, **kw):
    base = dict(features=[SingleBranchFeatureConfig(n, h) for n, h in features],
                single_branch_hidden_layers=[8], preference_hidden_layers=[], common_modality_dim=8,
                single_branch_input_dropout=None)
    base.update(kw)
    return SingleBranchNetEntityConfig(**base)


def cfg_to_meta(c):
    if isinstance(c, FeatureModuleConfig):
        return {'feature_name': c.feature_name, 'embedding_dim': c.embedding_dim,
                'pre_embedding_layers': c.pre_embedding_layers, 'activation_fn': c.activation_fn}
    d = {k: getattr(c, k) for k in ['single_branch_hidden_layers', 'common_modality_dim', 'activation_fn',
                                    'sampling_seed', 'single_branch_input_dropout', 'aggregation_fn',
                                    'normalize_single_branch_input', 'central_modality',
                                    'regularization_temperature', 'regularization_weight',
                                    'apply_output_activation', 'apply_batch_normalization',
                                    'apply_batch_norm_every']}
    d['embedding_regularization_type'] = c.embedding_regularization_type.value
    d['features'] = [{'feature_name': f.feature_name, 'feature_hidden_layers': f.feature_hidden_layers}
                     for f in c.features]
    d['train_modalities'] = sorted(c.train_modalities) if c.train_modalities else None
    d['eval_modalities'] = sorted(c.eval_modalities) if c.eval_modalities else None
    return d


def randomize_bn(module):
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.5, 0.5)
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)


# ------------------------------------------------------------------------------------------------
# G3 SingleBranchNetEntity (item side)
# ------------------------------------------------------------------------------------------------
def g3_entity():
    arrays, meta = world_arrays(), {'cases': []}
    feats = [('interactions', []), ('genres', []), ('text', [10]), ('audio', [])]
    R = EmbeddingRegularizationType
    variants = []
    for reg, central in [(R.NoRegularization, None), (R.PairwiseSingle, None), (R.CentralModality, 'text')]:
        for agg in ['mean', 'max']:
            for norm in [False, True]:
                variants.append(dict(embedding_regularization_type=reg, central_modality=central,
                                     aggregation_fn=agg, normalize_single_branch_input=norm,
                                     regularization_temperature=0.5, regularization_weight=0.3))
    variants.append(dict(apply_batch_normalization=False, apply_output_activation=True, activation_fn='tanh'))
    variants.append(dict(apply_batch_norm_every=1, single_branch_hidden_layers=[8, 6], activation_fn='selu',
                         embedding_regularization_type=R.PairwiseSingle))
    variants.append(dict(apply_batch_norm_every=-1, eval_modalities={'genres', 'text'}))
    for vi, kw in enumerate(variants):
        ds = make_dataset()
        ds.item_features['interactions'] = Feature(FeatureDefinition('interactions', FeatureType.VECTOR),
                                                   ds.item_sampling_matrix_train)
        cfg = entity_cfg(feats, **kw)
        torch.manual_seed(100 + vi)
        ent = SingleBranchNetEntity('item', ds.item_features, cfg, shared_common_dim=8)
        randomize_bn(ent)
        name = f'v{vi}'
        arrays.update(sd2n(ent.state_dict(), f'{name}/sd0/'))
        ent.train()
        idx = torch.from_numpy(np.random.default_rng(vi).integers(0, I, size=(6, 4)))
        # record the sampling decision by wrapping the reference's own sampler
        rec = {}
        orig = ent._sample_modalities

        def wrapped(indices, _orig=orig, _rec=rec):
            m = _orig(indices)
            _rec['mods'] = m
            return m
        ent._sample_modalities = wrapped
        y = ent(idx)
        r = torch.randn_like(y)
        reg_loss = ent.get_and_reset_other_loss()['reg_loss']
        ((y * r).sum() + reg_loss.sum()).backward()
        arrays[f'{name}/idx'] = t2n(idx)
        arrays[f'{name}/mods'] = rec['mods'].astype('U16')
        arrays[f'{name}/y'] = t2n(y)
        arrays[f'{name}/r'] = t2n(r)
        arrays[f'{name}/reg_loss'] = t2n(reg_loss)
        for k, p in ent.named_parameters():
            arrays[f'{name}/g/{k}'] = t2n(p.grad)
        arrays.update(sd2n(ent.state_dict(), f'{name}/sd1/'))
        ent.eval()
        all_items = torch.arange(I)
        with torch.no_grad():
            ye = ent(all_items)
        arrays[f'{name}/y_eval'] = t2n(ye)
        others = None
        if cfg.central_modality is not None:
            others = list(set(list(ent.train_modalities)) - {cfg.central_modality})
        meta['cases'].append({'name': name, 'cfg': cfg_to_meta(cfg),
                              'train_order': list(ent.train_modalities), 'eval_order': list(ent.eval_modalities),
                              'central_others_order': others})
    save('g3_entity', arrays, meta)


# ------------------------------------------------------------------------------------------------
# G4 full net + losses + grads, G8 optimizer trajectories
# ------------------------------------------------------------------------------------------------
def make_net(user_kind, seed, item_kw=None):
    ds = make_dataset()
    item_kw = item_kw or {}
    item = entity_cfg([('interactions', []), ('genres', []), ('text', [])],
                      embedding_regularization_type=EmbeddingRegularizationType.PairwiseSingle,
                      regularization_weight=0.1, **item_kw)
    if user_kind == 'lookup':
        user = FeatureModuleConfig('user_embedding', -1)
    elif user_kind == 'linear':
        user = FeatureModuleConfig('interactions', -1)
    else:
        user = entity_cfg([('interactions', []), ('gender', []), ('age', [])], single_branch_hidden_layers=[])
    cfg = SingleBranchNetConfig(user=user, item=item, shared_common_dim=8)
    torch.manual_seed(seed)
    net = SingleBranchNet(cfg, ds)
    randomize_bn(net)
    return net, cfg, ds


def record_mods(net):
    rec = {}
    for side in ('user', 'item'):
        mod = getattr(net, f'{side}_embedding_module')
        if isinstance(mod, SingleBranchNetEntity):
            orig = mod._sample_modalities

            def wrapped(indices, _orig=orig, _side=side):
                m = _orig(indices)
                rec.setdefault(_side, []).append(m)
                return m
            mod._sample_modalities = wrapped
    return rec


def net_meta(net, cfg):
    m = {'shared_common_dim': cfg.shared_common_dim, 'user': cfg_to_meta(cfg.user), 'item': cfg_to_meta(cfg.item)}
    for side in ('user', 'item'):
        mod = getattr(net, f'{side}_embedding_module')
        if isinstance(mod, SingleBranchNetEntity):
            m[f'{side}_train_order'] = list(mod.train_modalities)
            m[f'{side}_eval_order'] = list(mod.eval_modalities)
    return m


def batch(seed, b=6, n_neg=3):
    rng = np.random.default_rng(seed)
    u = torch.from_numpy(rng.integers(0, U, size=b))
    i = torch.from_numpy(rng.integers(0, I, size=(b, 1 + n_neg)))
    labels = torch.zeros(b, 1 + n_neg, dtype=torch.float64)
    labels[:, 0] = 1.
    return u, i, labels


def g4_full_net():
    arrays, meta = world_arrays(), {'cases': []}
    losses = {
        'bce': lambda: RecBinaryCrossEntropy(n_items=I, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3),
        'bpr': lambda: RecBayesianPersonalizedRankingLoss(n_items=I, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3),
        'bpr_sum': lambda: RecBayesianPersonalizedRankingLoss(n_items=I, aggregator='sum', train_neg_strategy='uniform_recbole', neg_train=3),
        'ssm_uniform': lambda: RecSampledSoftmaxLoss(n_items=I, aggregator='mean', train_neg_strategy='uniform', neg_train=3),
        'ssm_recbole': lambda: RecSampledSoftmaxLoss(n_items=I, aggregator='sum', train_neg_strategy='uniform_recbole', neg_train=3),
    }
    for ui, user_kind in enumerate(['lookup', 'linear', 'entity']):
        for li, (lname, lfn) in enumerate(losses.items()):
            name = f'{user_kind}_{lname}'
            net, cfg, ds = make_net(user_kind, 200 + ui)
            rec = record_mods(net)
            net.train()
            u, i, labels = batch(ui * 10 + li)
            arrays.update(sd2n(net.state_dict(), f'{name}/sd0/'))
            logits = net(u, i)
            arrays[f'{name}/logits'] = t2n(logits)          # before the in-place shift of sampled-softmax
            loss = lfn().compute_loss(logits, labels)
            reg = net.get_and_reset_other_loss()
            total = loss + reg['reg_loss']
            total.backward()
            arrays[f'{name}/u'] = t2n(u)
            arrays[f'{name}/i'] = t2n(i)
            arrays[f'{name}/labels'] = t2n(labels)
            arrays[f'{name}/rec_loss'] = t2n(loss)
            arrays[f'{name}/reg_loss'] = t2n(reg['reg_loss'])
            for side, lst in rec.items():
                arrays[f'{name}/{side}_mods'] = lst[0].astype('U16')
            for k, p in net.named_parameters():
                arrays[f'{name}/g/{k}'] = t2n(p.grad)
            arrays.update(sd2n(net.state_dict(), f'{name}/sd1/'))
            m = net_meta(net, cfg)
            m.update({'name': name, 'loss': lname, 'user_kind': user_kind,
                      'rec_loss_dtype': str(loss.dtype), 'logits_dtype': str(logits.dtype)})
            meta['cases'].append(m)
    save('g4_full_net', arrays, meta)


def g8_optim():
    arrays, meta = world_arrays(), {'cases': []}
    for oi, opt_name in enumerate(['adamw', 'adam', 'adagrad']):
        for user_kind in ['lookup', 'entity']:
            name = f'{opt_name}_{user_kind}'
            net, cfg, ds = make_net(user_kind, 300 + oi)
            rec = record_mods(net)
            net.train()
            opt = {'adam': torch.optim.Adam, 'adagrad': torch.optim.Adagrad,
                   'adamw': torch.optim.AdamW}[opt_name](net.parameters(), lr=1e-2, weight_decay=1e-2)
            loss_fn = RecBayesianPersonalizedRankingLoss(n_items=I, aggregator='mean',
                                                         train_neg_strategy='uniform_recbole', neg_train=3)
            arrays.update(sd2n(net.state_dict(), f'{name}/sd0/'))
            for s in range(3):
                u, i, labels = batch(1000 + s)
                logits = net(u, i)
                loss = loss_fn.compute_loss(logits, labels)
                reg = net.get_and_reset_other_loss()
                (loss + reg['reg_loss']).backward()
                opt.step()
                opt.zero_grad()
                arrays[f'{name}/u{s}'] = t2n(u)
                arrays[f'{name}/i{s}'] = t2n(i)
                arrays[f'{name}/labels{s}'] = t2n(labels)
                arrays[f'{name}/loss{s}'] = t2n(loss)
                arrays[f'{name}/reg{s}'] = t2n(reg['reg_loss'])
            for side, lst in rec.items():
                for s, mm in enumerate(lst):
                    arrays[f'{name}/{side}_mods{s}'] = mm.astype('U16')
            arrays.update(sd2n(net.state_dict(), f'{name}/sd3/'))
            m = net_meta(net, cfg)
            m.update({'name': name, 'optimizer': opt_name, 'lr': 1e-2, 'wd': 1e-2, 'user_kind': user_kind})
            meta['cases'].append(m)
    save('g8_optim', arrays, meta)


# ------------------------------------------------------------------------------------------------
# G5 InfoNCE
# ------------------------------------------------------------------------------------------------
def g5_infonce():
    arrays, meta = {}, {'cases': []}
    torch.manual_seed(5)
    for shape in [(6, 8), (5, 4, 8)]:
        for tau in [1.0, 0.1]:
            for red in ['mean', 'sum']:
                name = f'd{len(shape)}_t{tau}_{red}'
                a = torch.randn(*shape, requires_grad=True)
                b = torch.randn(*shape, requires_grad=True)
                loss = InfoNCE(tau, red)(a, b)
                loss.backward()
                arrays[f'{name}/a'], arrays[f'{name}/b'] = t2n(a), t2n(b)
                arrays[f'{name}/loss'] = t2n(loss)
                arrays[f'{name}/ga'], arrays[f'{name}/gb'] = t2n(a.grad), t2n(b.grad)
                meta['cases'].append({'name': name, 'tau': tau, 'reduction': red})
    save('g5_infonce', arrays, meta)


# ------------------------------------------------------------------------------------------------
# G6 negative sampling streams, G7 modality sampling streams
# ------------------------------------------------------------------------------------------------
class TinyTrainSet(torch.utils.data.Dataset):
    """Just the attributes the reference loaders read (data/dataset.py:325-396)."""

    def __init__(self, inter: sp.csr_matrix, n_neg, strategy):
        coo = inter.tocoo()
        self.rows, self.cols = coo.row, coo.col
        self.n_users, self.n_items = inter.shape
        self.items_in_split = np.arange(self.n_items)
        self.user_sampling_matrix = inter
        self.sampling_row_indices = [inter[i].indices for i in range(self.n_users)]
        self.n_negative_samples = n_neg
        self.negative_sampling_strategy = strategy
        pop = np.array(inter.sum(axis=0)).flatten()
        self.pop_distribution = pop / pop.sum()

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, k):
        return self.rows[k].astype('int64'), self.cols[k].astype('int64'), 1.


def g6_neg_sampling():
    arrays, meta = world_arrays(), {}
    ts = TinyTrainSet(W.inter, 3, 'uniform_recbole')
    arrays['coo_row'], arrays['coo_col'] = ts.rows.astype(np.int64), ts.cols.astype(np.int64)
    reproducible(42)
    loader = NegativeSamplingDataLoader(ts, batch_size=16, shuffle=True, num_workers=0)
    for b, (u, i, l) in enumerate(loader):
        if b >= 3:
            break
        arrays[f'recbole/u{b}'], arrays[f'recbole/i{b}'], arrays[f'recbole/l{b}'] = t2n(u), t2n(i), t2n(l)
    reproducible(42)
    arrays['randperm'] = t2n(torch.randperm(len(ts)))
    reproducible(42)
    sampler = NegativeSampler(ts, n_neg=3, neg_sampling_strategy='uniform')
    loader = TrainDataLoader(sampler, ts, batch_size=16, shuffle=True, num_workers=0, prefetch_factor=None)
    for b, (u, i, l) in enumerate(loader):
        if b >= 3:
            break
        arrays[f'uniform/u{b}'], arrays[f'uniform/i{b}'], arrays[f'uniform/l{b}'] = t2n(u), t2n(i), t2n(l)
    np.random.seed(42)
    rows = []
    for u in range(10):
        rows.append(ref_sampling.negative_sample_uniform(ts.items_in_split, 3, ts.sampling_row_indices[u]))
    arrays['dataset_uniform'] = np.stack(rows).astype(np.int64)
    meta.update({'batch_size': 16, 'n_neg': 3, 'seed': 42})
    save('g6_neg_sampling', arrays, meta)


def g7_row_wise_sample():
    arrays, meta = {}, {'cases': []}
    a3 = ['interactions', 'genres', 'text']
    a4 = ['interactions', 'genres', 'text', 'audio']
    for name, a, size, k, central in [('k1_n3', a3, (7, 5), 1, None), ('k2_n3', a3, (7, 5), 2, None),
                                      ('k2_n4', a4, (300,), 2, None), ('k1_n4', a4, (300,), 1, None),
                                      ('k2_n2', a3[:2], (64,), 2, None),
                                      ('central_n4', a4, (9, 4), 2, 'text')]:
        rng = np.random.default_rng(42)
        outs = []
        for rep in range(2):       # two consecutive calls: the generator state carries over
            outs.append(row_wise_sample(a, size, k=k, central_item=central, rng=rng).astype('U16'))
        arrays[f'{name}/call0'], arrays[f'{name}/call1'] = outs
        others = list(set(a) - {central}) if central is not None else None
        meta['cases'].append({'name': name, 'a': a, 'size': list(size), 'k': k, 'central': central,
                              'central_others_order': others})
    save('g7_row_wise_sample', arrays, meta)


# ------------------------------------------------------------------------------------------------
# G9 eval
# ------------------------------------------------------------------------------------------------
def g9_eval():
    arrays, meta = world_arrays(), {}
    net, cfg, ds = make_net('lookup', 900)
    net.eval()
    rng = np.random.default_rng(9)
    labels = (rng.random((U, I)) < 0.1) & ~(W.inter.toarray() > 0)
    with torch.no_grad():
        i_repr = net.get_item_representations(torch.arange(I))
        u_idx = torch.arange(U)
        u_repr = net.get_user_representations(u_idx)
        out = net.combine_user_item_representations(u_repr, i_repr)
        mask = torch.tensor(W.inter[u_idx.numpy()].toarray(), dtype=torch.bool)
        out[mask] = -torch.inf
    y = torch.from_numpy(labels.astype(np.float32))
    arrays.update(sd2n(net.state_dict(), 'sd/'))
    arrays['i_repr'], arrays['u_repr'], arrays['scores'] = t2n(i_repr), t2n(u_repr), t2n(out)
    arrays['labels'] = labels
    tk = torch.topk(out, 20, largest=True, sorted=True)
    arrays['topk_idx'], arrays['topk_val'] = t2n(tk.indices), t2n(tk.values)
    for k in [1, 10, 20]:
        arrays[f'ndcg@{k}'] = t2n(ref_metrics.ndcg_at_k_batch(out, y, k, aggr_sum=False))
        arrays[f'recall@{k}'] = t2n(ref_metrics.recall_at_k_batch(out, y, k, aggr_sum=False))
        arrays[f'precision@{k}'] = t2n(ref_metrics.precision_at_k_batch(out, y, k, aggr_sum=False))
    meta.update(net_meta(net, cfg))
    save('g9_eval', arrays, meta)


def g11_sgd_baseline():
    arrays = {}
    torch.manual_seed(11)
    m = SGDBaseline(U, I)
    with torch.no_grad():
        m.global_bias.fill_(0.25)
    u, i, _ = batch(11)
    arrays.update(sd2n(m.state_dict(), 'sd/'))
    arrays['u'], arrays['i'] = t2n(u), t2n(i)
    arrays['logits'] = t2n(m(u, i))
    save('g11_sgd_baseline', arrays, {})


if __name__ == '__main__':
    g1_polylinear()
    g2_feature_embedding()
    g3_entity()
    g4_full_net()
    g5_infonce()
    g6_neg_sampling()
    g7_row_wise_sample()
    g8_optim()
    g9_eval()
    g11_sgd_baseline()
    MANIFEST['_generator'] = {'torch': torch.__version__, 'numpy': np.__version__, 'PYTHONHASHSEED': '0',
                              'reference': 'Tigxy/SiBraR---Single-Branch-Recommender @ 2024-11-01'}
    with open(os.path.join(HERE, 'manifest.json'), 'w') as fh:
        json.dump(MANIFEST, fh, indent=1, sort_keys=True)
    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith('.npz'))
    print(f'wrote {len(MANIFEST) - 1} fixture files, {total / 1e6:.2f} MB')
