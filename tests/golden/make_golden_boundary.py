#!/usr/bin/env python3
"""Boundary evidence: the product's device-side containers fed with the REFERENCE's own objects.

    PYTHONHASHSEED=0 python tests/golden/make_golden_boundary.py            (build container only: imports /root/reference)

INTEGRATION.md claims that ``features.DeviceTable`` / ``evaluation._csr_to_device`` / ``FullEvaluator`` duck-type the reference's
``data.Feature.Feature`` and ``data.dataset.FullEvalDataset``. This script constructs the real objects — every feature kind the
path consumes (categorical, discrete, tag, vector, the CSR ``interactions`` VECTOR feature and the ``arange`` CATEGORICAL id
feature that SingleBranchNet adds itself, algorithms/sgd_alg.py:2021-2032, 2051-2059; sparse id sets of a cold-start split) and
the evaluation datasets of the two committed split directories — hands THEM to the product classes and records every buffer the
engine would upload (g16_boundary.npz). tests/test_host_cpu.py builds the same buffers from the product's own loader
(sibrar_amd.load_split_dataset -> HostFeature) and compares. Data only: no reference source travels.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden_split as mgs  # noqa: E402  (installs the import stubs, checks PYTHONHASHSEED)

from data import dataset as ref_dataset  # noqa: E402
from data.Feature import Feature  # noqa: E402
from data.config_classes import FeatureDefinition, FeatureType, InteractionDatasetConfig, TrainDatasetConfig  # noqa: E402

import sibrar_amd as S  # noqa: E402
from importlib import import_module  # noqa: E402

evaluation = import_module('sibrar---single-branch-recommender_amd.evaluation')


def table_buffers(prefix, feature, arrays, meta):
    t = S.DeviceTable(feature)                      # the product container, built from the reference's Feature object
    m = {'kind': t.kind, 'dim': int(t.dim), 'n_rows': int(t.n_rows)}
    for name in ('values', 'tags', 'indptr', 'indices', 'data', 'rowmap'):
        buf = getattr(t, name, None)
        if buf is not None:
            arrays[f'{prefix}/{name}'] = buf.numpy()
    for name in ('n_categories', 'pad', 'T', 'binary'):
        if hasattr(t, name):
            m[name] = int(getattr(t, name))
    meta[prefix] = m


def main():
    mgs.patch_config_loader()
    arrays, meta = {}, {}
    for name in ('split_random', 'split_cold_item'):
        d = os.path.join(HERE, name)
        common = dict(dataset_path=d, user_feature_definitions=mgs.USER_FEATS, item_feature_definitions=mgs.ITEM_FEATS,
                      model_requires_train_interactions=True, model_requires_item_interactions=True)
        tr = ref_dataset.TrainRecDataset(TrainDatasetConfig(split_set='train', n_negative_samples=3,
                                                            negative_sampling_strategy='uniform', **common))
        sets = {'train': tr}
        for s in ('val', 'test'):
            sets[s] = ref_dataset.FullEvalDataset(InteractionDatasetConfig(split_set=s, **common))
        for s, ds in sets.items():
            feats = {('user', k): f for k, f in ds.user_features.items()}
            feats.update({('item', k): f for k, f in ds.item_features.items()})
            # the features SingleBranchNet adds to the dataset's dictionaries (sgd_alg.py:2021-2032, 2051-2059)
            feats[('user', 'interactions')] = Feature(FeatureDefinition('interactions', FeatureType.VECTOR), raw_values=ds.user_sampling_matrix_train)
            feats[('user', 'user_embedding')] = Feature(FeatureDefinition('user_embedding', FeatureType.CATEGORICAL), raw_values=np.arange(ds.n_users))
            feats[('item', 'interactions')] = Feature(FeatureDefinition('interactions', FeatureType.VECTOR), raw_values=ds.item_sampling_matrix_train)
            feats[('item', 'item_embedding')] = Feature(FeatureDefinition('item_embedding', FeatureType.CATEGORICAL), raw_values=np.arange(ds.n_items))
            for (ent, k), f in feats.items():
                table_buffers(f'{name}/{s}/{ent}/{k}', f, arrays, meta)
            if s != 'train':
                # what evaluate_recommender_algorithm / FullEvaluator keep resident for the reference's FullEvalDataset
                ip, ix = evaluation._csr_to_device(ds.exclude_data, 'cpu')
                arrays[f'{name}/{s}/exclude/indptr'], arrays[f'{name}/{s}/exclude/indices'] = ip.numpy(), ix.numpy()
                lp, lx = S.FullEvaluator(dataset=ds)._labels('cpu')
                arrays[f'{name}/{s}/labels/indptr'], arrays[f'{name}/{s}/labels/indices'] = lp.numpy(), lx.numpy()
                arrays[f'{name}/{s}/items_in_split'] = np.asarray(ds.items_in_split).astype(np.int64)
                arrays[f'{name}/{s}/users_in_split'] = np.asarray(ds.users_in_split).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, 'g16_boundary.npz'), **arrays)
    with open(os.path.join(HERE, 'g16_boundary.json'), 'w') as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print('wrote', len(arrays), 'arrays,', len(meta), 'tables')


if __name__ == '__main__':
    main()
