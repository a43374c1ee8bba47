#!/usr/bin/env python3
"""G13: streams of the dataset-level negative samplers, generated with the REAL reference.

    python tests/golden/make_golden_samplers.py        (build container only: imports /root/reference/data/sampling.py)

``data/sampling.py`` is what ``TrainRecDataset._get_negative_samples`` (data/dataset.py:360-374) calls per interaction when
``use_dataset_negative_sampler`` is set: ``negative_sample_uniform`` (:18-32), ``negative_sample_uniform_recbole`` (:35-66),
``negative_sample_popular`` (:69-80). Inputs: the same tiny interaction world as the other fixtures (``golden_util.world``),
the global legacy numpy stream seeded with 42; ten users in a row per strategy, 3 negatives each. Only data is written.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _ref_import  # noqa: E402

_ref_import.install()
from data import sampling as ref_sampling  # noqa: E402
from golden_util import load, world  # noqa: E402

z = load('g6_neg_sampling')
w = world(z)
inter = w['inter']
n_users, n_items = inter.shape
positives = [inter[u].indices for u in range(n_users)]
pop = np.array(inter.sum(axis=0)).flatten()
pop = pop / pop.sum()
arrays = {'pop_distribution': pop}
# the split's item set: all items, and a cold-start-like subset with gaps (choices[neg] != neg)
subset = np.sort(np.random.default_rng(7).choice(n_items, size=30, replace=False))
for name, choices in (('all', np.arange(n_items)), ('subset', subset)):
    pos_in = [np.intersect1d(p, choices) for p in positives]
    np.random.seed(42)
    arrays[f'{name}/uniform'] = np.stack([ref_sampling.negative_sample_uniform(choices, 3, pos_in[u]) for u in range(10)]).astype(np.int64)
    arrays[f'{name}/after_uniform'] = np.random.randint(0, 1000, size=4)
    np.random.seed(42)
    arrays[f'{name}/uniform_recbole'] = np.stack([ref_sampling.negative_sample_uniform_recbole(choices, 3, pos_in[u])
                                                  for u in range(10)]).astype(np.int64)
    arrays[f'{name}/after_uniform_recbole'] = np.random.randint(0, 1000, size=4)
    for alpha in (1.0, 0.75):
        np.random.seed(42)
        arrays[f'{name}/popular_{alpha}'] = np.stack([ref_sampling.negative_sample_popular(choices, 3, pop, alpha, pos_in[u])
                                                      for u in range(10)]).astype(np.int64)
        arrays[f'{name}/after_popular_{alpha}'] = np.random.randint(0, 1000, size=4)
    arrays[f'{name}/choices'] = choices.astype(np.int64)
np.savez_compressed(os.path.join(HERE, 'g13_dataset_samplers.npz'), **arrays)
json.dump({'n_neg': 3, 'seed': 42, 'n_users_sampled': 10, 'alphas': [1.0, 0.75], 'choice_sets': ['all', 'subset']},
          open(os.path.join(HERE, 'g13_dataset_samplers.json'), 'w'), indent=1)
print({k: v.shape for k, v in arrays.items()})
