"""On-disk split format of the reference -> the dataset view the engine consumes.

A preprocessed dataset directory of the reference holds (data/data_preprocessing_utils.py:389-420, data/dataset.py:20-33):

    user_idxs.csv, item_idxs.csv                  at least the columns user_idx / item_idx (optional group_idx)
    listening_history_{train,val,test}.csv        at least user_idx, item_idx
    {user,item}_features_{split}.csv              tabular features: columns {entity}, {entity}_idx, one column per feature
    {user,item}_{feature}_{split}.npz             vector features: arrays ``indices`` [n], ``values`` [n, F]
    used_config.yaml                              the preprocessing config (``split.cold_start_type``)

``load_split_dataset`` follows RecDataset._load_data / _load_features (data/dataset.py:109-232), InteractionRecDataset
(:243-256), TrainRecDataset (:335-353) and FullEvalDataset (:399-438) and returns ONE object with the union of the attributes
those classes expose to the plugin (sgd_alg.py:2021-2067), the negative-sampling loader (data/dataloader.py:134-198) and the
evaluator (eval/eval.py:203-227). Feature processing follows data/Feature.py:193-288: categorical -> sorted unique values ->
integer ids; tags -> ``split(sep)`` -> sorted unique tags -> padded id matrix (pad id = number of tags); vector -> as stored.
Files are read with pandas / ``numpy.load(allow_pickle=False)`` / ``yaml.safe_load`` straight into host arrays; the engine
moves them to HBM once (features.DeviceTable).

Deliberate difference: the tags of one row are kept in sorted order (the reference iterates a Python ``set``, whose order
depends on PYTHONHASHSEED); the EmbeddingBag mean over them is order-independent up to fp32 rounding.
"""
from __future__ import annotations

import os
from ast import literal_eval
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import numpy as np
import scipy.sparse as sp

from .features import HostFeature

SPLIT_NAMES = ('train', 'val', 'test')
_MULTI_D = ('vector', 'matrix')


def _fdef(d) -> SimpleNamespace:
    """Feature definition from a dict / dataclass (config_classes.py:105-114): name, type, preprocessing, tag_split_sep."""
    get = (lambda k, default=None: d.get(k, default)) if isinstance(d, dict) else (lambda k, default=None: getattr(d, k, default))
    t = get('type')
    t = str(getattr(t, 'value', t)).lower()
    pre = get('preprocessing', 'none')
    pre = str(getattr(pre, 'value', pre) or 'none').lower()
    return SimpleNamespace(name=get('name'), type=t, preprocessing=pre, tag_split_sep=get('tag_split_sep'))


def _read_table(path: str, usecols):
    import pandas as pd
    return pd.read_csv(path, usecols=lambda c: c in usecols)


def _load_split_features(data_path: str, entity: str, fdefs: Sequence[SimpleNamespace], split: str):
    """data_preprocessing_utils.load_features :423-462 for one split -> (table | None, {name: (indices, values)})."""
    multi = [f.name for f in fdefs if f.type in _MULTI_D]
    tab = [f.name for f in fdefs if f.name not in multi]
    table = None
    if tab:
        path = os.path.join(data_path, f'{entity}_features_{split}.csv')
        if not os.path.exists(path):
            raise FileNotFoundError(f'Feature file "{path}" does not exist')
        table = _read_table(path, [entity, f'{entity}_idx'] + tab)
        missing = set(tab) - set(table.columns)
        if missing:
            raise ValueError(f'Column(s) for {entity} feature(s) {sorted(missing)} are missing.')
    md = {}
    for name in multi:
        path = os.path.join(data_path, f'{entity}_{name}_{split}.npz')
        if not os.path.exists(path):
            raise FileNotFoundError(f'Data file for {entity} feature "{name}" does not exist.')
        z = np.load(path, allow_pickle=False)
        idx, val = z['indices'], z['values']
        if len(idx) != len(val):
            raise ValueError(f'Mismatch between number of {entity} indices and its "{name}" feature'
                             f'({len(idx)} indices but {len(val)} feature values).')
        md[name] = (np.asarray(idx), np.asarray(val))
    return table, md


def _merge(entity: str, parts):
    """merge_features :470-509: rows of later splits whose index is not present yet are appended; result sorted by index."""
    import pandas as pd
    table, md = None, None
    for t, m in parts:
        if md is None:
            table, md = t, dict(m)
            continue
        if table is not None and t is not None:
            col = f'{entity}_idx'
            table = pd.concat([table, t[~t[col].isin(table[col])]])
        for k in md:
            ai, av = md[k]
            si, sv = m[k]
            new = np.isin(si, ai, assume_unique=True, invert=True)
            md[k] = (np.concatenate([ai, si[new]], axis=0), np.concatenate([av, sv[new]], axis=0))
    if table is not None:
        table = table.sort_values(f'{entity}_idx', kind='stable').reset_index(drop=True)
    if md:
        for k, (i, v) in md.items():
            o = np.argsort(i, kind='stable')
            md[k] = (i[o], v[o])
    return table, md or {}


def _tags_of(value, sep) -> List[str]:
    return sorted(set(str(value).split(sep)))


def build_feature(fd: SimpleNamespace, raw_values, indices, reference_values=None) -> HostFeature:
    """data/Feature.py:70-87 + :193-288 -> HostFeature (the processed values the engine keeps resident)."""
    raw = list(raw_values) if not hasattr(raw_values, 'shape') else raw_values
    indices = np.asarray(indices)
    if fd.type == 'categorical':
        uniq = set(raw)
        if reference_values is not None:
            uniq |= set(reference_values)
        uniq = sorted(tuple(uniq))
        vmap = {v: i for i, v in enumerate(uniq)}
        ids = np.array([vmap[v] for v in raw], dtype=np.int64)
        if fd.preprocessing == 'one_hot':
            return HostFeature(fd.name, 'dense', np.eye(len(uniq), dtype=np.float32)[ids], indices)
        return HostFeature(fd.name, 'categorical', ids, indices, n_categories=len(uniq), unique_values=uniq)
    if fd.type == 'tag':
        if fd.tag_split_sep is None:
            raise ValueError(f'For tag feature "{fd.name}" a separator (tag_split_sep) for the individual has to be provided. '
                             f'For genre tags "action|romance" this would be "|".')
        rows = [_tags_of(v, fd.tag_split_sep) for v in raw]
        uniq = set().union(*rows) if rows else set()
        if reference_values is not None:
            uniq |= set().union(*[_tags_of(v, fd.tag_split_sep) for v in reference_values])
        uniq = sorted(tuple(uniq))
        vmap = {v: i for i, v in enumerate(uniq)}
        lists = [[vmap[t] for t in r] for r in rows]
        width = max(map(len, lists)) if lists else 0
        padded = np.array([li + [len(uniq)] * (width - len(li)) for li in lists], dtype=np.int64).reshape(len(lists), width)
        if fd.preprocessing == 'multi_hot':
            hot = np.zeros((len(lists), len(uniq)), dtype=np.float32)
            for r, li in enumerate(lists):
                hot[r, li] = 1.
            return HostFeature(fd.name, 'dense', hot, indices)
        return HostFeature(fd.name, 'tag', padded, indices, n_categories=len(uniq))
    if fd.type == 'sequence':
        return HostFeature(fd.name, 'dense', np.stack([literal_eval(v) for v in raw], axis=0).astype(np.float32), indices)
    if fd.type in ('discrete', 'continuous'):
        return HostFeature(fd.name, 'dense', np.asarray(raw, dtype=np.float32).reshape(-1, 1), indices)
    if fd.type in _MULTI_D:
        v = np.stack(raw, axis=0) if isinstance(raw, list) else np.asarray(raw)
        if v.ndim != 2:
            raise NotImplementedError(f'feature "{fd.name}": only [n, F] vector features are on the SingleBranchNet path')
        return HostFeature(fd.name, 'dense', v, indices)
    raise ValueError(f'unknown feature type {fd.type!r}')


class SplitDataset:
    """Union of the RecDataset / InteractionRecDataset / TrainRecDataset / FullEvalDataset attributes the hot path reads."""

    def __init__(self, data_path: str, split_set: str = 'train', user_feature_definitions=None, item_feature_definitions=None,
                 n_negative_samples: int = 4, negative_sampling_strategy: str = 'uniform_recbole'):
        import pandas as pd
        import yaml
        if split_set not in SPLIT_NAMES:
            raise AssertionError(f'<{split_set}> is not a valid value for split set!')
        self.data_path, self.split_set = data_path, split_set
        self.is_train_split, self.is_eval_split = split_set == 'train', split_set in ('val', 'test')
        with open(os.path.join(data_path, 'used_config.yaml'), 'r') as fh:
            pre = yaml.safe_load(fh) or {}
        cst = str(((pre.get('split') or {}).get('cold_start_type')) or 'none').lower().split('.')[-1]
        self.cold_start_type = cst
        self.is_cold_start_user, self.is_cold_start_item = cst in ('user', 'both'), cst in ('item', 'both')
        self.is_cold_start_dataset = self.is_cold_start_user or self.is_cold_start_item

        user_idxs = pd.read_csv(os.path.join(data_path, 'user_idxs.csv'))
        item_idxs = pd.read_csv(os.path.join(data_path, 'item_idxs.csv'))
        self.n_users, self.n_items = len(user_idxs), len(item_idxs)
        self.n_user_groups, self.user_to_user_group = 0, None
        if 'group_idx' in user_idxs.columns:
            self.user_to_user_group = user_idxs[['user_idx', 'group_idx']].set_index('user_idx').sort_index().group_idx.to_numpy()
            self.n_user_groups = int(user_idxs.group_idx.nunique())

        lhs = self._history(split_set)
        if self.is_cold_start_dataset:
            self.users_in_split = np.array(sorted(lhs['user_idx'].unique()))
            self.items_in_split = np.array(sorted(lhs['item_idx'].unique()))
        else:
            self.users_in_split = user_idxs['user_idx'].to_numpy()
            self.items_in_split = item_idxs['item_idx'].to_numpy()
        self.n_interactions = len(lhs)
        self.n_users_in_split, self.n_items_in_split = len(self.users_in_split), len(self.items_in_split)
        self.interaction_matrix = self._matrix(lhs)
        train_lhs = lhs if self.is_train_split else self._history('train')
        if self.is_cold_start_dataset:
            self.train_users = np.array(sorted(train_lhs['user_idx'].unique()))
            self.train_items = np.array(sorted(train_lhs['item_idx'].unique()))
        else:
            self.train_users, self.train_items = user_idxs['user_idx'].to_numpy(), item_idxs['item_idx'].to_numpy()
        self.n_train_users, self.n_train_items = len(self.train_users), len(self.train_items)
        self.interaction_matrix_train = self._matrix(train_lhs)
        self.user_sampling_matrix = sp.csr_matrix(self.interaction_matrix)
        self.user_sampling_matrix_train = sp.csr_matrix(self.interaction_matrix_train)
        self.item_sampling_matrix_train = sp.csr_matrix(self.interaction_matrix_train.T)

        self.user_feature_definitions = [_fdef(d) for d in (user_feature_definitions or [])]
        self.item_feature_definitions = [_fdef(d) for d in (item_feature_definitions or [])]
        self.user_feature_names = [f.name for f in self.user_feature_definitions]
        self.item_feature_names = [f.name for f in self.item_feature_definitions]
        self.user_features = self._features('user', self.user_feature_definitions)
        self.item_features = self._features('item', self.item_feature_definitions)
        self.features = {'user': self.user_features, 'item': self.item_features}
        self.feature_names = {'user': self.user_feature_names, 'item': self.item_feature_names}

        self.n_negative_samples = n_negative_samples
        self.negative_sampling_strategy = negative_sampling_strategy
        self.exclude_data = self._interacted_mask()

    # ---- pieces -----------------------------------------------------------------------------------------------------------------
    def _history(self, split):
        return _read_table(os.path.join(self.data_path, f'listening_history_{split}.csv'), ['user_idx', 'item_idx'])

    def _matrix(self, lhs, dtype=np.int8):
        """dataset.py:159-176: one entry per history row (duplicates add up when converted to CSR, as in the reference)."""
        return sp.coo_matrix((np.ones(len(lhs), dtype=dtype), (lhs['user_idx'].to_numpy(), lhs['item_idx'].to_numpy())),
                             shape=(self.n_users, self.n_items))

    def _features(self, entity: str, fdefs) -> Dict[str, HostFeature]:
        """dataset.py:192-232: values of the split (+ 'val' when training), categories / tags from all three splits."""
        if not fdefs:
            return {}
        all_table, _ = _merge(entity, [_load_split_features(self.data_path, entity, fdefs, s) for s in SPLIT_NAMES])
        splits = (self.split_set, 'val') if self.is_train_split else (self.split_set,)
        table, md = _merge(entity, [_load_split_features(self.data_path, entity, fdefs, s) for s in splits])
        out = {}
        for fd in fdefs:
            if fd.type in _MULTI_D:
                idx, val = md[fd.name]
                out[fd.name] = build_feature(fd, val, idx)
            else:
                out[fd.name] = build_feature(fd, table[fd.name].tolist(), table[f'{entity}_idx'].to_numpy(),
                                             reference_values=all_table[fd.name].tolist())
        return out

    def _interacted_mask(self):
        """FullEvalDataset._get_interacted_mask (dataset.py:416-438): nothing for train, train for val, train + val for test."""
        mask = sp.csr_matrix(self.user_sampling_matrix_train.shape, dtype=bool)
        if self.split_set != 'train':
            mask = mask + self.user_sampling_matrix_train.astype(bool)
        if self.split_set == 'test':
            mask = mask + sp.csr_matrix(self._matrix(self._history('val'), dtype=bool))
        return sp.csr_matrix(mask)[:, self.items_in_split].astype(bool)

    def __len__(self):
        return self.n_interactions if self.is_train_split else self.n_users_in_split

    def eval_view(self):
        """The object itself already carries the FullEvalDataset attributes (labels = user_sampling_matrix of the split)."""
        return self


def load_split_dataset(data_path: str, split_set: str = 'train', user_feature_definitions=None, item_feature_definitions=None,
                       n_negative_samples: int = 4, negative_sampling_strategy: str = 'uniform_recbole') -> SplitDataset:
    return SplitDataset(data_path, split_set, user_feature_definitions, item_feature_definitions, n_negative_samples,
                        negative_sampling_strategy)
