#!/usr/bin/env python3
"""Golden fixture for the on-disk split loader, generated with the REAL reference in the build container:

    PYTHONHASHSEED=0 python tests/golden/make_golden_split.py

1. writes two tiny dataset directories in the reference's on-disk format (tests/golden/split_random, split_cold_item: CSV /
   NPZ / YAML data files only — the layout documented in data/dataset.py:20-33 and written by
   data/data_preprocessing_utils.py:389-420);
2. loads them with the reference's own TrainRecDataset / FullEvalDataset (data/dataset.py) and records what those objects
   expose (matrices, split index sets, processed feature values, exclusion masks) in g12_split_dataset.npz.
tests/test_host_cpu.py loads the same directories with sibrar_amd.load_split_dataset and compares.
The only patch applied to the reference: RecDataset._load_preprocessing_config builds its dataclass by hand (the YAML
deserialiser it uses, mashumaro, is not installed here).
"""
import json
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402,F401
import yaml  # noqa: E402

from data import dataset as ref_dataset  # noqa: E402
from data.config_classes import TrainDatasetConfig, InteractionDatasetConfig, FeatureDefinition, FeatureType  # noqa: E402
from data.preprocessing_config_classes import ColdStartType  # noqa: E402

assert os.environ.get('PYTHONHASHSEED') == '0', 'run with PYTHONHASHSEED=0'

U, I = 12, 9
USER_FEATS = [FeatureDefinition('gender', FeatureType.CATEGORICAL), FeatureDefinition('age', FeatureType.DISCRETE)]
ITEM_FEATS = [FeatureDefinition('genres', FeatureType.TAG, tag_split_sep='|'), FeatureDefinition('text', FeatureType.VECTOR)]


def write_dir(name, cold_item):
    d = os.path.join(HERE, name)
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(7 if cold_item else 3)
    pd.DataFrame({'user': [f'u{k}' for k in range(U)], 'user_idx': np.arange(U)}).to_csv(os.path.join(d, 'user_idxs.csv'), index=False)
    pd.DataFrame({'item': [f'i{k}' for k in range(I)], 'item_idx': np.arange(I)}).to_csv(os.path.join(d, 'item_idxs.csv'), index=False)
    item_sets = {'train': np.arange(0, 5), 'val': np.arange(5, 7), 'test': np.arange(7, 9)} if cold_item else \
        {s: np.arange(I) for s in ('train', 'val', 'test')}
    n_rows = {'train': 40, 'val': 14, 'test': 14}
    for s in ('train', 'val', 'test'):
        u = rng.integers(0, U, size=n_rows[s])
        i = rng.choice(item_sets[s], size=n_rows[s])
        lh = pd.DataFrame({'user_idx': u, 'item_idx': i, 'timestamp': np.arange(n_rows[s])}).drop_duplicates(['user_idx', 'item_idx'])
        lh.to_csv(os.path.join(d, f'listening_history_{s}.csv'), index=False)
    genders = np.array(['F', 'M', 'F', 'M', 'M', 'F', 'M', 'F', 'X', 'M', 'F', 'M'])     # 'X' appears only outside train
    ages = rng.integers(18, 60, size=U)
    # train + val together cover every user / item (a training run looks features up in their union, dataset.py:213-214)
    user_rows = {'train': np.arange(0, 8), 'val': np.arange(6, U), 'test': np.arange(0, U)}
    for s, rows in user_rows.items():
        pd.DataFrame({'user': [f'u{k}' for k in rows], 'user_idx': rows, 'gender': genders[rows], 'age': ages[rows],
                      'unused': 1}).to_csv(os.path.join(d, f'user_features_{s}.csv'), index=False)
    tags = ['rock|pop', 'jazz', 'pop', 'rock|metal|pop', 'folk', 'jazz|folk', 'metal', 'classical', 'pop|classical|jazz|rock']
    text = rng.standard_normal((I, 5)).astype(np.float32)
    for s in ('train', 'val', 'test'):
        rows = item_sets[s] if cold_item else {'train': np.arange(0, 7), 'val': np.array([8, 5, 6, 7]), 'test': np.arange(I)}[s]
        pd.DataFrame({'item': [f'i{k}' for k in rows], 'item_idx': rows, 'genres': [tags[k] for k in rows]}) \
            .to_csv(os.path.join(d, f'item_features_{s}.csv'), index=False)
        np.savez(os.path.join(d, f'item_text_{s}.npz'), indices=np.asarray(rows), values=text[rows])
    cfg = {'split': {'ratios': [0.8, 0.1, 0.1], 'split_type': 'cold_start' if cold_item else 'random',
                     'cold_start_type': 'item' if cold_item else None, 'seed': 42},
           'interactions': {}, 'user_features': [], 'item_features': []}
    with open(os.path.join(d, 'used_config.yaml'), 'w') as fh:
        yaml.safe_dump(cfg, fh)
    return d


def patch_config_loader():
    from types import SimpleNamespace

    def _load(self):
        with open(os.path.join(self.data_path, 'used_config.yaml')) as fh:
            c = yaml.safe_load(fh)
        cst = c['split'].get('cold_start_type')
        return SimpleNamespace(split=SimpleNamespace(cold_start_type=ColdStartType(cst) if cst else None))
    ref_dataset.RecDataset._load_preprocessing_config = _load


def csr(prefix, m):
    m = sp.csr_matrix(m)
    m.sum_duplicates()
    m.sort_indices()
    return {prefix + '/indptr': m.indptr.astype(np.int64), prefix + '/indices': m.indices.astype(np.int64),
            prefix + '/data': np.asarray(m.data).astype(np.int64), prefix + '/shape': np.array(m.shape, dtype=np.int64)}


def record(prefix, ds, arrays, meta):
    arrays[prefix + '/users_in_split'] = np.asarray(ds.users_in_split).astype(np.int64)
    arrays[prefix + '/items_in_split'] = np.asarray(ds.items_in_split).astype(np.int64)
    arrays.update(csr(prefix + '/interaction_matrix', ds.interaction_matrix))
    arrays.update(csr(prefix + '/user_sampling_matrix', ds.user_sampling_matrix))
    arrays.update(csr(prefix + '/user_sampling_matrix_train', ds.user_sampling_matrix_train))
    arrays.update(csr(prefix + '/item_sampling_matrix_train', ds.item_sampling_matrix_train))
    if hasattr(ds, 'exclude_data'):
        arrays.update(csr(prefix + '/exclude_data', ds.exclude_data))
    m = {'n_users': int(ds.n_users), 'n_items': int(ds.n_items), 'n_interactions': int(ds.n_interactions),
         'is_cold_start_user': bool(ds.is_cold_start_user), 'is_cold_start_item': bool(ds.is_cold_start_item), 'features': {}}
    for ent, feats in (('user', ds.user_features), ('item', ds.item_features)):
        for name, f in feats.items():
            key = f'{prefix}/{ent}/{name}'
            arrays[key + '/values'] = np.asarray(f.values)
            arrays[key + '/indices'] = np.asarray(f._indices).astype(np.int64)
            fm = {'type': str(f.feature_definition.type), 'dim': f.dim if not isinstance(f.dim, tuple) else list(f.dim)}
            if str(f.feature_definition.type) in ('categorical', 'tag'):
                fm['unique_values'] = [str(v) for v in f.unique_values]
            m['features'][f'{ent}/{name}'] = fm
    meta[prefix] = m


def main():
    patch_config_loader()
    arrays, meta = {}, {}
    for name, cold in (('split_random', False), ('split_cold_item', True)):
        d = write_dir(name, cold)
        common = dict(dataset_path=d, user_feature_definitions=USER_FEATS, item_feature_definitions=ITEM_FEATS,
                      model_requires_train_interactions=True, model_requires_item_interactions=True)
        tr = ref_dataset.TrainRecDataset(TrainDatasetConfig(split_set='train', n_negative_samples=3,
                                                            negative_sampling_strategy='uniform', **common))
        record(f'{name}/train', tr, arrays, meta)
        for s in ('val', 'test'):
            ev = ref_dataset.FullEvalDataset(InteractionDatasetConfig(split_set=s, **common))
            record(f'{name}/{s}', ev, arrays, meta)
    np.savez_compressed(os.path.join(HERE, 'g12_split_dataset.npz'), **arrays)
    with open(os.path.join(HERE, 'g12_split_dataset.json'), 'w') as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print('wrote', len(arrays), 'arrays')


if __name__ == '__main__':
    main()
