"""Helpers shared by the golden-vector tests: fixture loading and the synthetic 'world' the fixtures
were generated on (tests/golden/make_golden.py::make_world)."""
import json
import os

import numpy as np
import scipy.sparse as sp
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
U, I = 50, 40

with open(os.path.join(GOLDEN, 'manifest.json')) as fh:
    MANIFEST = json.load(fh)


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def sub(z, prefix):
    """All arrays under ``prefix`` as {key-without-prefix: tensor}."""
    return {k[len(prefix):]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(prefix)}


def state_dict(z, prefix, requires_grad=False):
    sd = {}
    for k, v in sub(z, prefix).items():
        v = v.clone()
        if requires_grad and v.dtype.is_floating_point and 'running_' not in k:
            v.requires_grad_(True)
        sd[k] = v
    return sd


def world(z):
    inter = sp.csr_matrix((np.ones(len(z['world/inter_indices']), dtype=np.int8),
                           z['world/inter_indices'], z['world/inter_indptr']), shape=(U, I))
    return {
        'inter': inter, 'inter_t': sp.csr_matrix(inter.T),
        'text': z['world/text'], 'audio': z['world/audio'],
        'genres': z['world/genres_padded'], 'genres_ntags': int(z['world/genres_ntags']),
        'gender': z['world/gender'], 'gender_ncat': int(z['world/gender_ncat']),
        'age': z['world/age'], 'age_ncat': int(z['world/age_ncat']),
    }


def ref_tables(w):
    """RefTables (oracle/model_ref.py) of the world, keyed like the reference's feature dicts
    (incl. the features SingleBranchNet adds itself, sgd_alg.py:2021-2059)."""
    from oracle.model_ref import RefTable
    user = {
        'gender': RefTable('categorical', w['gender'], n_categories=w['gender_ncat']),
        'age': RefTable('categorical', w['age'], n_categories=w['age_ncat']),
        'interactions': RefTable('csr', w['inter']),
        'user_embedding': RefTable('categorical', np.arange(U), n_categories=U),
    }
    item = {
        'text': RefTable('dense', w['text']),
        'audio': RefTable('dense', w['audio']),
        'genres': RefTable('tag', w['genres'], n_categories=w['genres_ntags']),
        'interactions': RefTable('csr', w['inter_t']),
        'item_embedding': RefTable('categorical', np.arange(I), n_categories=I),
    }
    return user, item


def close(a, b, rtol=1e-4, atol=1e-6, what='', norm_rtol=1e-5, scale=0.):
    """|a-b| <= atol + rtol*|b| + norm_rtol*max|b| (the last term covers elements that are sums with
    cancellation, e.g. the mathematically-zero gradient of a bias in front of a BatchNorm)."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, f'{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}'
    err = (a - b).abs()
    tol = atol + rtol * b.abs() + norm_rtol * max(float(b.abs().max()) if b.numel() else 0., float(scale))
    bad = err > tol
    assert not bad.any(), f'{what}: max abs err {err.max().item():.3e} (ref max {b.abs().max().item():.3e}), {int(bad.sum())} bad'


def gscale(tensors):
    """Largest magnitude over a group of gradient tensors: the scale of the rounding noise that a
    mathematically-zero member of the group (bias in front of a BatchNorm) carries."""
    return max([float(torch.as_tensor(t).abs().max()) for t in tensors if torch.as_tensor(t).numel()] + [0.])


def bn_shadowed_biases(keys):
    """Linear biases that feed a BatchNorm directly: their true gradient is exactly zero (the BN subtracts the
    batch mean), what autograd returns is rounding noise, and Adam/AdamW turn that noise into +-lr steps. Their
    trajectories are therefore chaotic in ANY implementation (the reference included) and are excluded from
    optimizer-trajectory comparisons. Found structurally: ``...linear_i.bias`` with a sibling ``batch_norm_i`` /
    trailing ``batch_norm`` (PolyLinear) or a BatchNorm1d right after the PolyLinear in ``sb_net``."""
    import re
    keys = list(keys)
    out = set()
    for k in keys:
        m = re.match(r'(.*)layers\.linear_(\d+)\.bias$', k)
        if not m:
            continue
        pre, i = m.group(1), int(m.group(2))
        n_layers = 1 + max(int(re.match(r'.*linear_(\d+)\.bias$', q).group(1)) for q in keys
                           if q.startswith(pre + 'layers.linear_') and q.endswith('.bias'))
        # (the running mean of that BatchNorm absorbs the bias, so it drifts with it)
        if f'{pre}layers.batch_norm_{i}.weight' in keys:
            out.update([k, f'{pre}layers.batch_norm_{i}.running_mean'])
        if i == n_layers - 1 and f'{pre}layers.batch_norm.weight' in keys:
            out.update([k, f'{pre}layers.batch_norm.running_mean'])
        ms = re.match(r'(.*sb_net\.)(\d+)\.$', pre)
        if ms and i == n_layers - 1 and f'{ms.group(1)}{int(ms.group(2)) + 1}.running_mean' in keys:
            out.update([k, f'{ms.group(1)}{int(ms.group(2)) + 1}.running_mean'])
    return out


def host_dataset(w):
    """The golden 'world' as a dataset namespace for the product plugin (sibrar_amd.SingleBranchNet)."""
    from types import SimpleNamespace
    import sibrar_amd as S
    item = {
        'text': S.HostFeature('text', 'dense', w['text']),
        'audio': S.HostFeature('audio', 'dense', w['audio']),
        'genres': S.HostFeature('genres', 'tag', w['genres'], n_categories=w['genres_ntags']),
    }
    user = {
        'gender': S.HostFeature('gender', 'categorical', w['gender'], n_categories=w['gender_ncat']),
        'age': S.HostFeature('age', 'categorical', w['age'], n_categories=w['age_ncat']),
    }
    return SimpleNamespace(n_users=U, n_items=I, user_features=user, item_features=item,
                           user_sampling_matrix_train=w['inter'], item_sampling_matrix_train=w['inter_t'],
                           is_cold_start_user=False, is_cold_start_item=False)


def side_cfg(d):
    """Golden manifests do not record ``preference_hidden_layers`` (unused by the model, required by the config class)."""
    d = dict(d)
    if 'features' in d:
        d.setdefault('preference_hidden_layers', [])
    return d


def product_net(z, case, sd_prefix, device='cuda'):
    """Build sibrar_amd.SingleBranchNet from a golden case and load the reference's state_dict into it."""
    import sibrar_amd as S
    cfg = S.SingleBranchNetConfig.from_dict({'shared_common_dim': case['shared_common_dim'], 'user': side_cfg(case['user']),
                                             'item': side_cfg(case['item'])})
    orders = {k: case[k2] for k, k2 in [('user_train', 'user_train_order'), ('user_eval', 'user_eval_order'),
                                        ('item_train', 'item_train_order'), ('item_eval', 'item_eval_order')] if k2 in case}
    net = S.SingleBranchNet(cfg, host_dataset(world(z)), modality_orders=orders)
    missing = net.load_state_dict(state_dict(z, sd_prefix), strict=True)
    return net.to(device)
