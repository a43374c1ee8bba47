"""Oracle (test infrastructure): numpy restatement of the reference's sampling streams.

Integer / RNG work — the parity bar is bit-exact index streams.

Reference files followed (relative to /root/reference):
  utilities/utils.py:60-90        -> ``row_wise_sample``  (per-row Generator.choice, the slow literal form)
  data/dataloader.py:134-198      -> ``recbole_collate``  (default sampler (i), global legacy np.random)
  data/dataloader.py:57-58,93-131 -> ``uniform_collate``  (sampler (iii))
  data/sampling.py:7-32           -> ``dataset_uniform``  (sampler (ii))
  utilities/utils.py:22-27        -> ``reproducible``
"""
from __future__ import annotations

import math
import random

import numpy as np


def reproducible(seed: int):
    """utilities/utils.py:22-27."""
    import torch
    random.seed(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)


def loader_epoch_order(n: int) -> np.ndarray:
    """Index order of one epoch of ``DataLoader(dataset, shuffle=True, num_workers=0)`` as the reference
    builds it (data/data_utils.py:18-59 -> torch.utils.data): creating the loader iterator draws one int64
    ``_base_seed`` from torch's default CPU generator, then ``RandomSampler`` draws a second int64 to seed a
    private generator and takes ``torch.randperm(n, generator)`` from it."""
    import torch
    torch.empty((), dtype=torch.int64).random_()
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).numpy()


def row_wise_sample(a: list, size, k: int = 2, replace: bool = False, central_item=None, rng=None) -> np.ndarray:
    """utilities/utils.py:60-90: one ``choice(a, k, replace)`` call per output row, in row order.
    With a central item, column 0 is the central item and the remaining k-1 columns are drawn from
    the other items (whose order is ``a`` with the central item removed — the reference uses
    ``list(set(a) - {central})``, hash-order dependent; callers pass an explicit ``a``)."""
    if isinstance(size, int):
        size = (size,)
    size = tuple(size)
    if central_item is None:
        choose = rng.choice if rng is not None else np.random.choice
        n_rows = math.prod(size)
        rows = [choose(a, k, replace=replace) for _ in range(n_rows)]
        return np.array(rows).reshape(size + (k,))
    if central_item not in a:
        raise ValueError(f'central item "{central_item}" must be contained in "a"')
    others = [x for x in a if x != central_item]
    rest = row_wise_sample(others, size, k=k - 1, replace=replace, rng=rng)
    width = max(len(str(x)) for x in a)
    out = np.empty(size + (k,), dtype=f'<U{width}' if isinstance(a[0], str) else type(a[0]))
    out[..., 0] = central_item
    out[..., 1:] = rest
    return out


def recbole_collate(user_idx: np.ndarray, pos_item_idx: np.ndarray, n_neg: int, items_in_split: np.ndarray,
                    positives_of_user) -> tuple:
    """data/dataloader.py:154-198 (NegativeSamplingDataLoader._neg_sampling_collate_fn).

    All B*n_neg slots are drawn with ``np.random.choice(items_in_split, n, replace=True)`` from the global
    legacy RNG; only slots that hit one of the slot-user's positives are redrawn (same call, size = number
    of colliding slots) until none collide. Slot s belongs to user ``s % B`` (tile layout, :178) and the
    result is ``value_ids.reshape(n_neg, B).T`` (:192).  ``positives_of_user[u]`` is the array of items of
    user u in the split's interaction matrix (dataset.py:339)."""
    user_idx = np.asarray(user_idx).astype(np.int64)
    b = len(user_idx)
    total = b * n_neg
    values = np.zeros(total, dtype=np.int64)
    slot_user = np.tile(user_idx, n_neg)
    todo = np.arange(total)
    while len(todo) > 0:
        values[todo] = np.random.choice(items_in_split, size=len(todo), replace=True)
        keep = [s for s in todo if values[s] in positives_of_user[slot_user[s]]]
        todo = np.array(keep, dtype=np.int64)
    neg = values.reshape(n_neg, -1).T
    items = np.column_stack([pos_item_idx, neg]).astype(np.int64)
    labels = np.zeros_like(items, dtype=float)          # float64, dataloader.py:196
    n_pos = pos_item_idx.shape[-1] if np.ndim(pos_item_idx) > 1 else 1
    labels[:, :n_pos] = 1.
    return user_idx, items, labels


def uniform_collate(user_idx: np.ndarray, pos_item_idx: np.ndarray, n_neg: int, n_items: int,
                    positives_of_user) -> tuple:
    """data/dataloader.py:93-131 (TrainDataLoader + NegativeSampler 'uniform'): ``np.random.randint(0, n_items, m)``
    fills the masked slots in row-major order; a slot stays masked while its value is one of the row's
    user positives (np.isin)."""
    user_idx = np.asarray(user_idx).astype(np.int64)
    b = len(user_idx)
    neg = np.empty((b, n_neg), dtype=np.int64)
    mask = np.ones((b, n_neg), dtype=bool)
    m = int(mask.sum())
    while True:
        neg[mask] = np.random.randint(0, high=n_items, size=m)
        for r in range(b):
            mask[r] = np.isin(neg[r], positives_of_user[user_idx[r]])
        m = int(mask.sum())
        if m == 0:
            break
    items = np.column_stack([pos_item_idx, neg]).astype(np.int64)
    labels = np.zeros_like(items, dtype=float)
    n_pos = pos_item_idx.shape[-1] if np.ndim(pos_item_idx) > 1 else 1
    labels[:, :n_pos] = 1.
    return user_idx, items, labels


def dataset_uniform(choices: np.ndarray, size: int, positive_indices: np.ndarray) -> np.ndarray:
    """data/sampling.py:7-32 (negative_sample_uniform): draw ``size`` distinct ranks from the
    ``len(choices) - n_pos`` non-positive slots and shift each rank past the positives below it."""
    if len(choices) - len(positive_indices) < size:
        raise ValueError(f'Not enough values in the range to sample "{size}" unique values.')
    pos = np.searchsorted(choices, positive_indices)
    raw = np.random.choice(len(choices) - len(pos), size=size, replace=False)
    shifted = pos - np.arange(len(pos))
    return choices[raw + np.searchsorted(shifted, raw, side='right')]


def dataset_uniform_recbole(choices: np.ndarray, size: int, positive_indices: np.ndarray) -> np.ndarray:
    """data/sampling.py:35-66 (negative_sample_uniform_recbole): draw POSITIONS in ``choices`` with ``np.random.randint`` and
    redraw the ones that are ``in positive_indices`` — the reference tests the position itself against the positive item ids
    (identical to testing the item only when ``choices`` is ``arange``); restated literally."""
    n_choices, n_positive = len(choices), len(positive_indices)
    if n_choices - n_positive < size:
        raise ValueError(f'Not enough values in the range to sample "{size}" unique values.')
    if (n_choices - n_positive) * 0.5 < size:
        raise ValueError('Sampling is really inefficient either because the number of choices are small'
                         'or the number of items to sample is too high.')
    neg = np.full(size, fill_value=-1)
    todo = list(range(size))
    while len(todo):
        neg[todo] = np.random.randint(low=0, high=n_choices, size=len(todo))
        todo = [i for i, v in zip(todo, neg[todo]) if v in positive_indices]
    return choices[neg]


def dataset_popular(choices: np.ndarray, size: int, popularity_distribution: np.ndarray, squashing_factor: float,
                    positive_indices: np.ndarray = None) -> np.ndarray:
    """data/sampling.py:69-80 (negative_sample_popular): ``np.random.choice`` over the non-positive items with probabilities
    popularity ** alpha, renormalised (with replacement)."""
    if positive_indices is not None:
        choices = np.setdiff1d(choices, positive_indices, assume_unique=True)
    p = np.power(popularity_distribution[choices], squashing_factor)
    p = p / p.sum()
    return np.random.choice(choices, size=size, p=p)


def dataset_sampler_collate(user_idx, pos_item_idx, n_neg, strategy, choices, positives_of_user, pop=None, alpha=1.0):
    """TrainRecDataset.__getitem__ with use_dataset_negative_sampler (data/dataset.py:360-394) over the rows of one batch, in batch
    order, followed by the default collate: -> (u [B], items [B, 1 + n_neg], labels [B, 1 + n_neg] float64)."""
    items, labels = [], []
    for u, i in zip(user_idx, pos_item_idx):
        if strategy == 'uniform':
            neg = dataset_uniform(choices, n_neg, positives_of_user[u])
        elif strategy == 'uniform_recbole':
            neg = dataset_uniform_recbole(choices, n_neg, positives_of_user[u])
        elif strategy == 'popular':
            neg = dataset_popular(choices, n_neg, pop, alpha, positives_of_user[u])
        else:
            raise ValueError(f'Sampling strategy "{strategy}" not yet supported.')
        items.append(np.concatenate([[i], neg]))
        labels.append(np.concatenate([np.array([1.]), np.zeros_like(neg, dtype=float)]))
    return np.asarray(user_idx, dtype=np.int64), np.stack(items).astype(np.int64), np.stack(labels)
