"""Oracle (test infrastructure): CPU restatement of the SingleBranchNet model.

Plain PyTorch-CPU ops on a flat ``state_dict`` whose keys are the reference's
own (SURVEY.md §8(b) "state_dict contract"), so that a golden ``state_dict``
captured from the reference can be fed to this restatement and to the HIP
engine alike.  Functional style: nothing here is an ``nn.Module``.

Reference files followed (relative to /root/reference):
  modules/polylinear.py:17-77            -> ``poly_linear``
  algorithms/sgd_alg.py:1279-1396        -> ``feature_embedding``
  algorithms/sgd_alg.py:1764-2006        -> ``RefEntity``
  algorithms/sgd_alg.py:2009-2144        -> ``RefSingleBranchNet``
  algorithms/sgd_alg.py:88-123           -> ``sgd_baseline_logits``
  algorithms/sgd_alg.py:126-200          -> ``mf_logits``
  algorithms/sgd_alg.py:1399-1614        -> ``feature_mf_forward``
  algorithms/sgd_alg.py:1617-1762        -> ``dropoutnet_entity`` / ``dropoutnet_forward``
  data/Feature.py:140-162                -> ``RefTable.rows``
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # torch.nn.BatchNorm1d default, polylinear.py:61
BN_MOMENTUM = 0.1    # torch.nn.BatchNorm1d default


def _act(name: Optional[str], x: torch.Tensor) -> torch.Tensor:
    """modules/polylinear.py:5-10 (ACTIVATION_FN_MAP)."""
    if name is None:
        return x
    if name == 'relu':
        return torch.relu(x)
    if name == 'tanh':
        return torch.tanh(x)
    if name == 'sigmoid':
        return torch.sigmoid(x)
    if name == 'selu':
        return torch.selu(x)
    raise ValueError(f'unknown activation {name!r}')


def batch_norm(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, training: bool) -> torch.Tensor:
    """BatchNorm1d over the row dimension (polylinear.py:61,68; sgd_alg.py:1837).

    Train: batch mean / biased variance normalise; running stats are updated
    in ``sd`` with momentum 0.1 and the *unbiased* variance. Eval: running stats.
    """
    w, b = sd[prefix + 'weight'], sd[prefix + 'bias']
    rm, rv = sd[prefix + 'running_mean'], sd[prefix + 'running_var']
    if training:
        n = x.shape[0]
        mean = x.mean(dim=0)
        var = ((x - mean) ** 2).mean(dim=0)
        with torch.no_grad():
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            unbiased = var.detach() * (n / max(n - 1, 1))
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unbiased)
            if prefix + 'num_batches_tracked' in sd:
                sd[prefix + 'num_batches_tracked'] += 1
    else:
        mean, var = rm, rv
    return (x - mean) / torch.sqrt(var + BN_EPS) * w + b


def poly_linear(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, layer_config: Sequence[int],
                act: str, out_act: Optional[str], bn_every: int, training: bool) -> torch.Tensor:
    """modules/polylinear.py:18-77 — layer i is linear_i -> [batch_norm_i] -> act (i < n-1);
    then [batch_norm] when bn_every == -1; then the output activation."""
    n_layers = len(layer_config) - 1
    assert n_layers >= 1
    for i in range(n_layers):
        x = x @ sd[f'{prefix}layers.linear_{i}.weight'].t() + sd[f'{prefix}layers.linear_{i}.bias']
        if bn_every > 0 and (i + 1) % bn_every == 0:
            x = batch_norm(x, sd, f'{prefix}layers.batch_norm_{i}.', training)
        if i < n_layers - 1:
            x = _act(act, x)
    if bn_every == -1:
        x = batch_norm(x, sd, f'{prefix}layers.batch_norm.', training)
    return _act(out_act, x)


# ----------------------------------------------------------------------------------------------
# feature tables (host views of data/Feature.py objects)
# ----------------------------------------------------------------------------------------------
class RefTable:
    """Host view of one reference ``Feature`` (data/Feature.py:27-295): the processed values
    plus the global-index -> row map (Feature.py:50-51, 146)."""

    def __init__(self, kind: str, values, indices: Optional[np.ndarray] = None, n_categories: int = None):
        assert kind in ('dense', 'csr', 'categorical', 'tag')
        self.kind = kind
        self.values = values
        n = values.shape[0]
        self.indices = np.arange(n) if indices is None else np.asarray(indices)
        self.row_of = {int(g): r for r, g in enumerate(self.indices)}
        if kind in ('dense', 'csr'):
            self.dim = int(values.shape[1])
        elif kind == 'tag':
            self.dim = int(n_categories)             # number of distinct tags; pad id == dim
        else:
            self.dim = 0
        self.n_categories = n_categories

    def rows(self, idx) -> np.ndarray:
        """Feature.__getitem__ (Feature.py:140-157): flat global ids -> stored rows."""
        flat = np.asarray(idx).reshape(-1)
        return np.fromiter((self.row_of[int(g)] for g in flat), dtype=np.int64, count=flat.size)

    def fetch(self, idx) -> np.ndarray:
        r = self.rows(idx)
        v = self.values[r]
        if self.kind == 'csr':
            v = v.toarray()          # Feature.py:149-150 densifies sparse rows
        return v


def table_from_feature(feature) -> RefTable:
    """Duck-typed conversion of a reference-like Feature (``feature_definition.type``,
    ``values``, ``_indices``) into a RefTable."""
    import scipy.sparse as sp
    ftype = str(getattr(feature.feature_definition.type, 'value', feature.feature_definition.type)).lower()
    idx = getattr(feature, '_indices', None)
    if ftype == 'categorical':
        return RefTable('categorical', np.asarray(feature.values), idx, n_categories=feature.n_unique_categories)
    if ftype == 'tag':
        return RefTable('tag', np.asarray(feature.values), idx, n_categories=int(feature.dim))
    if sp.issparse(feature.values):
        return RefTable('csr', sp.csr_matrix(feature.values), idx)
    v = np.asarray(feature.values)
    if v.ndim == 1:
        v = v[:, None]
    return RefTable('dense', v, idx)


def feature_embedding(sd: Dict[str, torch.Tensor], prefix: str, table: RefTable, idx: torch.Tensor,
                      embedding_dim: Optional[int], hidden: Optional[Sequence[int]], act: str,
                      training: bool) -> torch.Tensor:
    """FeatureEmbedding.forward (sgd_alg.py:1373-1389) for flat ``idx`` [n] -> [n, out]."""
    flat = idx.reshape(-1).cpu().numpy()
    if table.kind in ('dense', 'csr'):
        x = torch.from_numpy(np.ascontiguousarray(table.fetch(flat))).float()        # sgd_alg.py:1380 x.float()
        cfg = [table.dim] + list(hidden or []) + ([embedding_dim] if embedding_dim is not None else [])
        if len(cfg) > 1:
            # sgd_alg.py:1356: output_fn == activation_fn (activation also on the projector output)
            x = poly_linear(x, sd, prefix + 'pre_embedding_layers.', cfg, act, act, 0, training)
        return x
    w = sd[prefix + 'embedding_layer.weight']
    vals = torch.from_numpy(np.ascontiguousarray(table.fetch(flat))).long()
    if table.kind == 'categorical':
        return w[vals]                                                                   # sgd_alg.py:1331,1386
    # TAG: EmbeddingBag(n_tags + 1, C, padding_idx=-1), default mode 'mean' (sgd_alg.py:1336-1337):
    # mean over the non-padding tags of each bag; an all-padding bag gives zeros.
    pad = w.shape[0] - 1
    keep = (vals != pad)
    summed = (w[vals] * keep.unsqueeze(-1).to(w.dtype)).sum(dim=-2)
    cnt = keep.sum(dim=-1).clamp(min=1).to(w.dtype)
    return summed / cnt.unsqueeze(-1)


def _cfg(obj, name, default=None):
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def _reg_type(cfg) -> str:
    t = _cfg(cfg, 'embedding_regularization_type', 'no_regularization')
    return str(getattr(t, 'value', t))


def info_nce(a: torch.Tensor, b: torch.Tensor, temperature: float = 1., reduction: str = 'mean') -> torch.Tensor:
    """train/regularization_losses.py:14-43 — symmetric cross entropy of a @ b^T / tau against the diagonal."""
    logits = (a @ b.transpose(-2, -1)) / temperature
    n = logits.shape[-1]
    rows = logits.reshape(-1, n)
    cols = logits.transpose(-2, -1).reshape(-1, n)
    target = torch.arange(n).repeat(rows.shape[0] // n)
    return F.cross_entropy(rows, target, reduction=reduction) + F.cross_entropy(cols, target, reduction=reduction)


class RefEntity:
    """SingleBranchNetEntity (sgd_alg.py:1764-2006) on a flat state_dict.

    ``modality_order`` replaces the reference's ``list(set_of_names)`` (sgd_alg.py:1907,1931) whose
    order depends on PYTHONHASHSEED; goldens record the order the reference used.
    """

    def __init__(self, sd, prefix: str, cfg, tables: Dict[str, RefTable], shared_dim: int,
                 val_interactions_available: bool = True, train_order: Optional[List[str]] = None,
                 eval_order: Optional[List[str]] = None):
        self.sd, self.prefix, self.cfg, self.tables, self.D = sd, prefix, cfg, tables, shared_dim
        feats = _cfg(cfg, 'features')
        if len(feats) == 0:
            raise ValueError('SingleBranchEntity requires at least one feature.')       # sgd_alg.py:1775-1776
        names = [_cfg(f, 'feature_name') for f in feats]
        self.hidden = {_cfg(f, 'feature_name'): (_cfg(f, 'feature_hidden_layers') or []) for f in feats}
        tm = _cfg(cfg, 'train_modalities')
        train = set(tm) if tm else set(names)                                            # sgd_alg.py:1880-1882
        em = _cfg(cfg, 'eval_modalities')
        if em is not None:
            for m in em:
                if m not in train:
                    raise ValueError(f'Cannot use modality "{m}" during evaluation, if it is not used during training.')
        ev = set(em) if em else set(train)                                               # sgd_alg.py:1894
        if not val_interactions_available:
            ev.discard('interactions')                                                   # sgd_alg.py:1896-1897
        if not train or not ev:
            raise ValueError('No single modality is available')
        missing = train - set(tables.keys())
        if missing:
            raise ValueError(f'Features for modalities {missing} are not available!')
        self.train_modalities = list(train_order) if train_order is not None else [n for n in names if n in train]
        self.eval_modalities = list(eval_order) if eval_order is not None else [n for n in names if n in ev]
        assert set(self.train_modalities) == train and set(self.eval_modalities) == ev
        self.C = _cfg(cfg, 'common_modality_dim')
        self.act = _cfg(cfg, 'activation_fn', 'relu')
        self.H = list(_cfg(cfg, 'single_branch_hidden_layers') or [])
        self.dropout = _cfg(cfg, 'single_branch_input_dropout')
        self.normalize = bool(_cfg(cfg, 'normalize_single_branch_input', False))
        self.agg = _cfg(cfg, 'aggregation_fn', 'mean')
        if self.agg not in ('mean', 'max'):
            raise ValueError(f'Aggregation function "{self.agg}" is not supported.')
        self.reg_type = _reg_type(cfg)
        self.central = _cfg(cfg, 'central_modality')
        self.tau = _cfg(cfg, 'regularization_temperature', 1.)
        self.reg_weight = _cfg(cfg, 'regularization_weight', 1.)
        self.out_act = self.act if _cfg(cfg, 'apply_output_activation', False) else None
        bn = bool(_cfg(cfg, 'apply_batch_normalization', True))
        every = _cfg(cfg, 'apply_batch_norm_every', 0)
        self.bn_every = every if bn else 0                                               # sgd_alg.py:1819
        self.trailing_bn = bn and every == 0                                             # sgd_alg.py:1834-1837
        self.p = 1 if self.dropout is not None else 0     # index of PolyLinear inside nn.Sequential
        self.rng = np.random.default_rng(_cfg(cfg, 'sampling_seed', 42))                 # sgd_alg.py:1848
        self.reg_loss = torch.zeros(1)

    # -- sampling (sgd_alg.py:1904-1932) ---------------------------------------------------
    def sample_modalities(self, shape, training: bool) -> np.ndarray:
        from .sampling_ref import row_wise_sample
        if not training:
            out = np.empty(tuple(shape) + (len(self.eval_modalities),), dtype=object)
            out[...] = self.eval_modalities
            return out
        a = list(self.train_modalities)
        if self.reg_type == 'no_regularization':
            return row_wise_sample(a, tuple(shape), k=1, rng=self.rng)
        if self.reg_type == 'pairwise_single':
            return row_wise_sample(a, tuple(shape), k=2, rng=self.rng)
        if self.reg_type == 'central_modality':
            return row_wise_sample(a, tuple(shape), k=2, central_item=self.central, rng=self.rng)
        raise ValueError(f'Embedding regularization "{self.reg_type}" is not yet supported.')

    # -- per-modality front ends + scatter (sgd_alg.py:1934-1978) -----------------------------
    def modality_embeddings(self, idx: torch.Tensor, mods: np.ndarray, training: bool) -> torch.Tensor:
        if tuple(idx.shape) != tuple(mods.shape[:-1]):
            raise ValueError('Shape of indices and modalities (up to the last dimension) does not match.')
        k = mods.shape[-1]
        flat_idx = torch.repeat_interleave(idx.reshape(-1), k)
        flat_mods = mods.reshape(-1)
        out = torch.zeros(flat_idx.numel(), self.C)
        parts, rows_of = [], []
        for m in sorted(set(flat_mods.tolist())):                                        # np.unique order
            rows = np.flatnonzero(flat_mods == m)
            e = feature_embedding(self.sd, f'{self.prefix}modality_modules.{m}.', self.tables[m],
                                  flat_idx[rows], self.C, self.hidden[m], self.act, training)
            parts.append(e.reshape(len(rows), self.C))
            rows_of.append(torch.from_numpy(rows))
        out = out.index_put((torch.cat(rows_of),), torch.cat(parts))
        return out.reshape(tuple(idx.shape) + (k, self.C))

    # -- shared single-branch MLP (sgd_alg.py:1865-1877) -------------------------------------
    def embed(self, idx: torch.Tensor, mods: np.ndarray, training: bool, dropout_mask=None) -> torch.Tensor:
        x = self.modality_embeddings(idx, mods, training)
        lead = x.shape[:-1]
        x = x.reshape(-1, self.C)
        if self.normalize:
            x = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)                        # F.normalize(p=2, eps=1e-12)
        if self.dropout is not None and training:
            if dropout_mask is not None:
                x = x * dropout_mask / (1 - self.dropout)
            else:
                x = F.dropout(x, self.dropout, True)
        x = poly_linear(x, self.sd, f'{self.prefix}sb_net.{self.p}.', [self.C] + self.H + [self.D],
                        self.act, self.out_act, self.bn_every, training)
        if self.trailing_bn:
            x = batch_norm(x, self.sd, f'{self.prefix}sb_net.{self.p + 1}.', training)
        return x.reshape(tuple(lead) + (self.D,))

    def forward(self, idx: torch.Tensor, training: bool, mods: Optional[np.ndarray] = None,
                dropout_mask=None) -> torch.Tensor:
        """sgd_alg.py:1850-1863. ``mods`` overrides the sampled modality array (recorded goldens)."""
        if mods is None:
            mods = self.sample_modalities(idx.shape, training)
        e = self.embed(idx, mods, training, dropout_mask)
        if training:
            if self.reg_type == 'no_regularization':
                self.reg_loss = torch.zeros(1)
            else:
                if e.shape[-2] != 2:
                    raise SystemError('second last dimension of embeddings should be of size 2')
                self.reg_loss = info_nce(e[..., 0, :], e[..., 1, :], self.tau)           # sgd_alg.py:1985-1989
        if self.agg == 'mean':
            return e.mean(dim=-2)
        return e.max(dim=-2).values

    def get_and_reset_other_loss(self):
        loss = self.reg_loss * self.reg_weight                                           # sgd_alg.py:2001-2006
        self.reg_loss = torch.zeros(1)
        return {'reg_loss': loss}


class RefSingleBranchNet:
    """SingleBranchNet (sgd_alg.py:2009-2144) on a flat state_dict. Each side is either a RefEntity or a
    plain feature front end (lookup or Linear-on-interactions; sgd_alg.py:2041-2046, 2068-2073)."""

    def __init__(self, sd, cfg, user_tables: Dict[str, RefTable], item_tables: Dict[str, RefTable],
                 is_cold_start_user=False, is_cold_start_item=False, orders: Optional[dict] = None):
        self.sd, self.cfg = sd, cfg
        self.D = _cfg(cfg, 'shared_common_dim')
        orders = orders or {}
        self.sides = {}
        self.tables = {'user': user_tables, 'item': item_tables}
        for side, tables, cold in (('user', user_tables, is_cold_start_user), ('item', item_tables, is_cold_start_item)):
            c = _cfg(cfg, side)
            if _cfg(c, 'features') is not None:
                self.sides[side] = RefEntity(sd, f'{side}_embedding_module.', c, tables, self.D,
                                             val_interactions_available=not cold,
                                             train_order=orders.get(f'{side}_train'),
                                             eval_order=orders.get(f'{side}_eval'))
            else:
                self.sides[side] = c

    def _repr(self, side: str, idx: torch.Tensor, training: bool, mods=None) -> torch.Tensor:
        s = self.sides[side]
        if isinstance(s, RefEntity):
            return s.forward(idx, training, mods)
        dim = _cfg(s, 'embedding_dim')
        dim = self.D if dim == -1 else dim                                               # sgd_alg.py:2043-2044
        e = feature_embedding(self.sd, f'{side}_embedding_module.', self.tables[side][_cfg(s, 'feature_name')],
                              idx, dim, _cfg(s, 'pre_embedding_layers'), _cfg(s, 'activation_fn', 'relu'), training)
        return e.reshape(tuple(idx.shape) + (-1,))

    def user_repr(self, u_idx, training, mods=None):
        return self._repr('user', u_idx, training, mods)

    def item_repr(self, i_idx, training, mods=None):
        return self._repr('item', i_idx, training, mods)

    @staticmethod
    def combine(u: torch.Tensor, i: torch.Tensor) -> torch.Tensor:
        """sgd_alg.py:2093-2114: all-pairs when the item side is 2-D, per-slot dot otherwise."""
        if i.dim() == 2:
            return u @ i.t()
        return (u.unsqueeze(1) * i).sum(-1)

    def forward(self, u_idx, i_idx, training=True, user_mods=None, item_mods=None):
        return self.combine(self.user_repr(u_idx, training, user_mods), self.item_repr(i_idx, training, item_mods))

    def get_and_reset_other_loss(self):
        """sgd_alg.py:2127-2140."""
        losses = {'reg_loss': torch.zeros(1)}
        for side in ('user', 'item'):
            s = self.sides[side]
            if isinstance(s, RefEntity):
                r = s.get_and_reset_other_loss()
                losses['reg_loss'] = losses['reg_loss'] + r['reg_loss']
                losses[f'{side}_reg_loss'] = r['reg_loss']
        return losses


def sgd_baseline_logits(sd, u_idx: torch.Tensor, i_idx: torch.Tensor) -> torch.Tensor:
    """SGDBaseline (sgd_alg.py:110-119): user_bias[u] + item_bias[i].squeeze() + global_bias."""
    return sd['user_bias.weight'][u_idx] + sd['item_bias.weight'][i_idx].squeeze() + sd['global_bias']


def mf_logits(sd, u_idx: torch.Tensor, i_idx: torch.Tensor) -> torch.Tensor:
    """SGDMatrixFactorization (sgd_alg.py:159-194): (u_embed[:, None, :] * i_embed).sum(-1) [+ item_bias[i].squeeze()]
    [+ global_bias]; ``i_idx`` [B, N] (training) or [I] (all-pairs, the evaluation path eval/eval.py:209-217). A ``user_bias``
    in the state dict raises like the reference's in-place add of a [B, 1, 1] tensor to [B, N] (:190)."""
    u_embed = sd['user_embeddings.weight'][u_idx]
    i_embed = sd['item_embeddings.weight'][i_idx]
    out = (u_embed[:, None, :] * i_embed).sum(dim=-1)
    if 'user_bias.weight' in sd:
        out += sd['user_bias.weight'][u_idx][:, None]
    if 'item_bias.weight' in sd:
        out = out + sd['item_bias.weight'][i_idx].squeeze()
    if 'global_bias' in sd:
        out = out + sd['global_bias']
    return out


def feature_mf_forward(sd, side: str, table: RefTable, u_idx: torch.Tensor, i_idx: torch.Tensor, *, embedding_dim: int,
                       intermediate_layers=None, aggregate_for_rec: bool = False, temperature: float = 0.1,
                       embedding_loss_aggregator: str = 'mean', training: bool = True):
    """ItemFeatureMatrixFactorization (side='item', sgd_alg.py:1399-1505) / UserFeatureMatrixFactorization (side='user',
    :1508-1614): -> (logits, contrastive loss). The content branch is ``embedding_net`` = FeatureEmbedding(feature,
    pre_embedding_layers=intermediate_layers, embedding_dim); the scores use the profile embedding or, with
    ``aggregate_for_rec``, the mean of profile and content; the loss is InfoNCE(profile, content) — for users on [B, 1, D]
    inputs (one pair per group: identically zero), returned unscaled (lambda_content is never applied)."""
    idx = i_idx if side == 'item' else u_idx
    profile = sd[f'{side}_embeddings.weight'][idx]
    content = feature_embedding(sd, 'embedding_net.', table, idx, embedding_dim, intermediate_layers, 'relu', training)
    content = content.reshape(*idx.shape, content.shape[-1])
    mixed = torch.stack([profile, content], dim=0).mean(dim=0) if aggregate_for_rec else profile
    if side == 'item':
        u_embed, i_embed = sd['user_embeddings.weight'][u_idx], mixed
        reg = info_nce(profile, content, temperature, embedding_loss_aggregator)
    else:
        u_embed, i_embed = mixed, sd['item_embeddings.weight'][i_idx]
        reg = info_nce(profile[:, None, :], content[:, None, :], temperature, embedding_loss_aggregator)
    out = (u_embed[:, None, :] * i_embed).sum(dim=-1)
    if 'item_bias.weight' in sd:
        out = out + sd['item_bias.weight'][i_idx].squeeze()
    if 'global_bias' in sd:
        out = out + sd['global_bias']
    return out, reg


def dropoutnet_entity(sd, prefix: str, ecfg: dict, tables: Dict[str, RefTable], idx: torch.Tensor, preferences: torch.Tensor,
                      shared_dim: int, training: bool) -> torch.Tensor:
    """DropoutNetEntity.forward (sgd_alg.py:1645-1655): pref_net(preferences) (PolyLinear defaults: ReLU between layers AND on the
    output), content modules, concat [*content, pref], net (PolyLinear with the entity's activation, ReLU on the output)."""
    pl = [preferences.shape[-1]] + list(ecfg['preference_layers'])
    pref = poly_linear(preferences.reshape(-1, pl[0]), sd, prefix + 'pref_net.', pl, 'relu', 'relu', 0, training)
    cont = []
    for j, f in enumerate(ecfg['features']):
        c = feature_embedding(sd, f'{prefix}cont_modules.{j}.', tables[f['feature_name']], idx, f.get('embedding_dim'),
                              f.get('pre_embedding_layers'), f.get('activation_fn', 'relu'), training)
        cont.append(c.reshape(-1, c.shape[-1]))
    x = torch.cat([*cont, pref], dim=-1)
    shape = [x.shape[-1]] + list(ecfg['common_hidden_layers']) + [shared_dim]
    y = poly_linear(x, sd, prefix + 'net.', shape, ecfg.get('activation_fn', 'relu'), 'relu', 0, training)
    return y.reshape(*idx.shape, shared_dim)


def dropoutnet_forward(sd, cfg: dict, user_tables, item_tables, inter, inter_t, u_idx: torch.Tensor, i_idx: torch.Tensor,
                       user_strategy=None, item_strategy=None, training: bool = True) -> torch.Tensor:
    """DropoutNet.forward (sgd_alg.py:1689-1757). ``inter`` / ``inter_t``: user x item and item x user CSR training matrices
    (dataset.get_{user,item}_interaction_vectors = matrix[idx].toarray(), data/dataset.py:306-319). strategies: 1 = Normal,
    2 = NoPreference (zero preference vector), one per user and one per ROW of ``i_idx``; None = all Normal (evaluation)."""
    def prefs(idx, matrix, strategy):
        flat = idx.reshape(-1).numpy()
        dense = torch.from_numpy(np.asarray(matrix[flat].toarray())).float().reshape(*idx.shape, matrix.shape[1])
        if strategy is not None:
            drop = torch.from_numpy(np.asarray(strategy) != 1)
            dense[drop] = 0.
        return dense
    u = dropoutnet_entity(sd, 'user_net.', cfg['user'], user_tables, u_idx, prefs(u_idx, inter, user_strategy),
                          cfg['shared_common_dim'], training)
    i = dropoutnet_entity(sd, 'item_net.', cfg['item'], item_tables, i_idx, prefs(i_idx, inter_t, item_strategy),
                          cfg['shared_common_dim'], training)
    if i.ndim == 2:
        return torch.einsum('be,ce->bc', u, i)
    return torch.einsum('be,bce->bc', u, i)


def init_state_dict(shapes: Dict[str, tuple], seed: int = 42) -> Dict[str, torch.Tensor]:
    """Initialise a state_dict by the reference's rules (train/utils.py:5-13):
    Linear kaiming-uniform(relu) bound sqrt(6/fan_in), bias 0; Embedding N(0, 0.1/dim); EmbeddingBag
    N(0,1) with zero pad row; BatchNorm weight 1 / bias 0 / mean 0 / var 1. ``shapes`` maps key -> (shape, kind)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, (shape, kind) in shapes.items():
        if kind == 'linear_w':
            bound = math.sqrt(6.0 / shape[1])
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == 'embedding':
            sd[key] = torch.randn(shape, generator=g) * (0.1 / shape[-1])
        elif kind == 'bag':
            w = torch.randn(shape, generator=g)
            w[-1] = 0
            sd[key] = w
        elif kind == 'ones':
            sd[key] = torch.ones(shape)
        elif kind == 'count':
            sd[key] = torch.zeros((), dtype=torch.long)
        else:
            sd[key] = torch.zeros(shape)
    return sd
