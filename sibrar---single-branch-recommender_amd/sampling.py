"""Host-side samplers that reproduce the reference's random streams bit-exactly, vectorised per batch.

* ``sample_modality_ids`` replaces the per-row ``rng.choice(a, k, replace=False)`` loop of
  utilities/utils.py:60-90 (row_wise_sample), called from algorithms/sgd_alg.py:1904-1927 with the entity's
  ``np.random.default_rng(sampling_seed)`` (sgd_alg.py:1848). For the small populations used here numpy's
  ``Generator.choice(n, k, replace=False)`` is Floyd's algorithm followed by a Fisher-Yates pass over the k results, every
  bounded draw being one 32-bit Lemire draw from the PCG64 stream; one vectorised ``rng.integers`` call with per-element
  bounds consumes the stream in exactly the same order.
* ``recbole_negative_collate`` replaces data/dataloader.py:154-198 (the default ``uniform_recbole`` collate): identical
  ``np.random.choice(items_in_split, n, replace=True)`` calls on the global legacy generator, with the Python
  ``v in positives`` loop replaced by a sorted-key membership test.
* ``loader_epoch_order`` reproduces the index order of ``DataLoader(shuffle=True)``.
All functions return integer arrays; the parity bar is exact equality with the oracle / golden streams.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np
import torch


def sample_modality_ids(rng: np.random.Generator, n_rows: int, n_mod: int, k: int) -> np.ndarray:
    """-> int8 [n_rows, k] of positions into the ordered modality list; same stream consumption as n_rows calls of
    ``rng.choice(n_mod, k, replace=False)``."""
    if k > n_mod:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    if k == 1:
        return rng.integers(0, n_mod, size=(n_rows, 1)).astype(np.int8)
    if k == 2:
        hi = np.tile(np.array([n_mod - 1, n_mod, 2], dtype=np.int64), n_rows)
        w = rng.integers(0, hi).reshape(n_rows, 3)
        v0 = w[:, 0]
        v1 = np.where(w[:, 1] == v0, n_mod - 1, w[:, 1])          # Floyd: a repeated value is replaced by j = n-1
        swap = w[:, 2] == 0                                        # Fisher-Yates step i=1: swap with position 0
        first = np.where(swap, v1, v0)
        second = np.where(swap, v0, v1)
        return np.stack([first, second], axis=1).astype(np.int8)
    # general (unused by the shipped configs): literal per-row calls
    return np.stack([rng.choice(n_mod, k, replace=False) for _ in range(n_rows)]).astype(np.int8)


def sample_modalities(rng, order: Sequence[str], n_rows: int, reg_type: str, central: Optional[str] = None) -> np.ndarray:
    """Modality positions (into ``order``) for ``n_rows`` index slots — sgd_alg.py:1912-1927."""
    n = len(order)
    if reg_type == 'no_regularization':
        return sample_modality_ids(rng, n_rows, n, 1)
    if reg_type == 'pairwise_single':
        return sample_modality_ids(rng, n_rows, n, 2)
    if reg_type == 'central_modality':
        if central not in order:
            raise ValueError(f'central item "{central}" must be contained in "a"')
        c = list(order).index(central)
        others = np.array([i for i in range(n) if i != c], dtype=np.int8)
        pick = sample_modality_ids(rng, n_rows, n - 1, 1)[:, 0]
        return np.stack([np.full(n_rows, c, dtype=np.int8), others[pick]], axis=1)
    raise ValueError(f'Embedding regularization "{reg_type}" is not yet supported.')


def loader_epoch_order(n: int) -> np.ndarray:
    """Index order of one epoch of ``DataLoader(shuffle=True, num_workers=0)``: creating the iterator draws the int64
    ``_base_seed`` from torch's default generator, RandomSampler draws another int64 to seed its private generator, then
    ``torch.randperm(n, generator)``."""
    torch.empty((), dtype=torch.int64).random_()
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).numpy()


class PositiveIndex:
    """Sorted (user, item) keys of the split's interaction matrix for O(log nnz) membership tests."""

    def __init__(self, csr):
        csr = csr.tocsr()
        self.n_items = int(csr.shape[1])
        rows = np.repeat(np.arange(csr.shape[0], dtype=np.int64), np.diff(csr.indptr))
        self.keys = np.sort(rows * self.n_items + csr.indices.astype(np.int64))
        csr.sort_indices()
        self._h_indptr = np.ascontiguousarray(csr.indptr, dtype=np.int64)       # for the native small-batch collate
        self._h_indices = np.ascontiguousarray(csr.indices, dtype=np.int32)
    HOST_BELOW = 4096

    def contains(self, users: np.ndarray, items: np.ndarray) -> np.ndarray:
        q = users.astype(np.int64) * self.n_items + items.astype(np.int64)
        pos = np.searchsorted(self.keys, q)
        pos = np.minimum(pos, len(self.keys) - 1) if len(self.keys) else pos
        return (self.keys[pos] == q) if len(self.keys) else np.zeros(len(q), dtype=bool)


class DevicePositiveIndex:
    """The same membership test on the GPU: the split's interaction CSR is resident in HBM, each round uploads the
    still-colliding (user, item) pairs, a kernel binary-searches the users' rows and the flags come back. It runs on a
    private stream so that it never waits for queued training kernels. The random draws stay on the host (bit-exact
    streams); only the `v in positives` test (data/dataloader.py:184-191) moves."""

    def __init__(self, csr, device):
        import torch
        from . import ops  # noqa: F401  (fails loudly when the HIP library is missing)
        csr = csr.tocsr()
        csr.sort_indices()
        self.device = torch.device(device)
        self.indptr = torch.from_numpy(csr.indptr.astype(np.int64)).to(self.device)
        self.indices = torch.from_numpy(csr.indices.astype(np.int32)).to(self.device)
        # high priority: the 10-us test kernel must not queue behind the training step's long kernels (the collate thread
        # blocks on its result two or three times per batch)
        self.stream = torch.cuda.Stream(device=self.device, priority=-1)
        self._h_indptr = np.ascontiguousarray(csr.indptr, dtype=np.int64)       # host copy for the small redraw rounds
        self._h_indices = np.ascontiguousarray(csr.indices, dtype=np.int32)
        # a cache-resident CSR (ML-1M: 2.6 MB of column indices) answers ~20 queries per microsecond on the host, a large one
        # (c2: 20 MB) pays a cache miss or two per query
        self.HOST_BELOW = 4096 if csr.nnz <= 2_000_000 else 512

    HOST_BELOW = 1024        # queries: below this a GPU round trip (copy + launch + sync, ~80 us) loses to the host search

    def contains(self, users: np.ndarray, items: np.ndarray) -> np.ndarray:
        import torch
        from ._lib import call, lib, ptr
        n = len(users)
        if n == 0:
            return np.zeros(0, dtype=bool)
        if n < self.HOST_BELOW:
            # the redraw rounds of a collate shrink geometrically (a few dozen pairs after the first): per-row binary search on
            # a host copy of the CSR (native, csrc/host_rng.hip)
            u = np.ascontiguousarray(users, dtype=np.int64)
            v = np.ascontiguousarray(items, dtype=np.int64)
            out = np.empty(n, dtype=np.uint8)
            rc = lib().sbr_host_csr_contains(self._h_indptr.ctypes.data, self._h_indices.ctypes.data, u.ctypes.data, v.ctypes.data,
                                             n, out.ctypes.data)
            if rc != 0:
                raise RuntimeError(lib().sbr_last_error().decode())
            return out.astype(bool)
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            # one pageable -> device copy for both operands (synchronous, see _lib.to_device)
            q = torch.from_numpy(np.concatenate([np.asarray(users, dtype=np.int64), np.asarray(items, dtype=np.int64)])).to(self.device)
            out = torch.empty(n, dtype=torch.uint8, device=self.device)
            call('sbr_csr_contains', ptr(self.indptr), ptr(self.indices), q.data_ptr(), q.data_ptr() + 8 * n, n, ptr(out),
                 self.stream.cuda_stream)
            res = out.cpu()
        return res.numpy().astype(bool)


def recbole_negative_collate(user_idx: np.ndarray, pos_item_idx: np.ndarray, n_neg: int, items_in_split: np.ndarray,
                             positives: PositiveIndex, identity: Optional[bool] = None):
    """data/dataloader.py:154-198 with the same global-RNG calls: draw all B*n_neg slots, redraw only the colliding ones
    (in ascending slot order) until none collides. Returns (users i64 [B], items i64 [B, 1+n_neg], labels f64).
    ``identity``: whether items_in_split is 0..n-1 (computed when not given; loaders pass it once)."""
    user_idx = np.asarray(user_idx).astype(np.int64)
    b = len(user_idx)
    total = b * n_neg
    n_cand = len(items_in_split)
    if identity is None:
        identity = is_arange(items_in_split)
    h_indptr = getattr(positives, '_h_indptr', None)
    # a slot's user is user_idx[slot % B]: the B rows of the CSR that a batch touches become cache-resident after the first round,
    # whatever the size of the CSR, so the host search is cheap for small batches even on large interaction matrices
    if (h_indptr is not None and 0 < total <= 8192 and np.ndim(pos_item_idx) == 1
            and 1 <= n_cand <= 0xFFFFFFFF and os.environ.get('SBR_NATIVE_COLLATE', '1') != '0'):
        return _recbole_collate_native(user_idx, pos_item_idx, n_neg, items_in_split, identity, positives)

    def draw(m):
        # np.random.choice(arr, m, replace=True) == arr[np.random.randint(0, len(arr), m)] on the legacy global stream
        # (checked in tests/test_host_cpu.py); the lookup is skipped when arr is 0..n-1
        r = legacy_randint(n_cand, m)
        return r if identity else items_in_split[r]

    slot_user = np.tile(user_idx, n_neg)
    values = draw(total).astype(np.int64, copy=False)
    todo = np.flatnonzero(positives.contains(slot_user, values))
    while len(todo) > 0:
        values[todo] = draw(len(todo))
        todo = todo[positives.contains(slot_user[todo], values[todo])]
    items = np.empty((b, 1 + n_neg) if np.ndim(pos_item_idx) == 1 else (b, pos_item_idx.shape[-1] + n_neg), dtype=np.int64)
    n_pos = items.shape[1] - n_neg
    if n_pos == 1 and values.flags.c_contiguous:
        from ._lib import lib
        pos_c = np.ascontiguousarray(pos_item_idx, dtype=np.int64)
        if lib().sbr_host_assemble_items(pos_c.ctypes.data, values.ctypes.data, b, n_neg, items.ctypes.data) != 0:
            raise RuntimeError(lib().sbr_last_error().decode())
    else:
        items[:, :n_pos] = np.asarray(pos_item_idx).reshape(b, n_pos)
        items[:, n_pos:] = values.reshape(n_neg, b).T
    return user_idx, items, _first_columns_positive(items.shape, n_pos)


_LABELS = {}


def _first_columns_positive(shape, n_pos):
    """The label matrix of the collates (first n_pos columns 1, negatives 0, float64 — data/dataloader.py:196-197): the same for
    every batch of a shape, so one array per shape is handed out instead of a fresh 0.7 MB one per batch (consumers treat labels as read-only; a writable
    array keeps torch.from_numpy quiet)."""
    key = (tuple(shape), n_pos)
    lab = _LABELS.get(key)
    if lab is None:
        lab = np.zeros(shape, dtype=float)
        lab[:, :n_pos] = 1.
        if len(_LABELS) < 64:
            _LABELS[key] = lab
    return lab


def _recbole_collate_native(user_idx, pos_item_idx, n_neg, items_in_split, identity, positives):
    """Small batches (the reference's default 256 x 10 slots): the whole collate in one native call (csrc/host_rng.hip,
    sbr_host_recbole_collate) on the host copy of the interaction CSR — same draws from the global legacy stream, same
    generator state afterwards as the numpy formulation above (tests/test_host_cpu.py)."""
    import ctypes
    from ._lib import lib
    b = len(user_idx)
    st = np.random.get_state()
    key = np.array(st[1], dtype=np.uint32)
    pos = ctypes.c_int(int(st[2]))
    pos_items = np.ascontiguousarray(pos_item_idx, dtype=np.int64)
    split = None if identity else np.ascontiguousarray(items_in_split, dtype=np.int64)
    items = np.empty((b, 1 + n_neg), dtype=np.int64)
    scratch = np.empty((2, b * n_neg), dtype=np.int64)
    rc = lib().sbr_host_recbole_collate(key.ctypes.data, ctypes.byref(pos), user_idx.ctypes.data, pos_items.ctypes.data, b, n_neg,
                                        len(items_in_split), None if split is None else split.ctypes.data,
                                        positives._h_indptr.ctypes.data, positives._h_indices.ctypes.data, items.ctypes.data,
                                        scratch[0].ctypes.data, scratch[1].ctypes.data)
    if rc != 0:
        raise RuntimeError(lib().sbr_last_error().decode())
    np.random.set_state((st[0], key, pos.value, st[3], st[4]))
    labels = np.zeros(items.shape, dtype=float)
    labels[:, 0] = 1.
    return user_idx, items, labels


def legacy_randint(high: int, n: int) -> np.ndarray:
    """``np.random.randint(0, high, size=n)`` on the global legacy generator — same values, same final generator state.
    Large draws run in the native replica (csrc/host_rng.hip, branch-free acceptance, ~2.3x numpy's speed): the state is taken
    from numpy, advanced natively and handed back."""
    if n < 2048 or high - 1 > 0xFFFFFFFF or high < 1:
        return np.random.randint(0, high, size=n)
    import ctypes
    from ._lib import lib
    st = np.random.get_state()
    key = np.array(st[1], dtype=np.uint32)                 # private copy, advanced in place
    pos = ctypes.c_int(int(st[2]))
    out = np.empty(n, dtype=np.int64)
    rc = lib().sbr_host_mt19937_randint(key.ctypes.data, ctypes.byref(pos), int(high), int(n), out.ctypes.data)
    if rc != 0:
        raise RuntimeError(lib().sbr_last_error().decode())
    np.random.set_state((st[0], key, pos.value, st[3], st[4]))
    return out


def is_arange(a: np.ndarray) -> bool:
    """True when ``a`` is the identity map 0..n-1 (items_in_split of a split that keeps every item)."""
    n = len(a)
    return bool(n == 0 or (a[0] == 0 and a[-1] == n - 1 and np.array_equal(a, np.arange(n))))


def uniform_negative_collate(user_idx: np.ndarray, pos_item_idx: np.ndarray, n_neg: int, n_items: int,
                             positives: PositiveIndex):
    """data/dataloader.py:93-131 (TrainDataLoader + NegativeSampler 'uniform'): ``np.random.randint(0, n_items, m)`` refills
    the still-colliding slots (row-major order) until none collides."""
    user_idx = np.asarray(user_idx).astype(np.int64)
    b = len(user_idx)
    neg = np.empty((b, n_neg), dtype=np.int64)
    mask = np.ones((b, n_neg), dtype=bool)
    users2d = np.repeat(user_idx[:, None], n_neg, axis=1)
    while True:
        m = int(mask.sum())
        if m == 0:
            break
        neg[mask] = np.random.randint(0, high=n_items, size=m)
        mask = positives.contains(users2d.reshape(-1), neg.reshape(-1)).reshape(b, n_neg)
    items = np.column_stack([pos_item_idx, neg]).astype(np.int64)
    labels = np.zeros_like(items, dtype=float)
    n_pos = pos_item_idx.shape[-1] if np.ndim(pos_item_idx) > 1 else 1
    labels[:, :n_pos] = 1.
    return user_idx, items, labels


# ---- dataset-level negative samplers (TrainRecDataset._get_negative_samples, data/dataset.py:360-374; data/sampling.py) ---------
# Called per interaction when ``use_dataset_negative_sampler`` is set. Host-side index work on the global legacy numpy stream:
# the results and the stream position afterwards are bit-identical to the reference's functions (tests/golden g13).
def dataset_negative_uniform(choices: np.ndarray, size: int, positives: np.ndarray) -> np.ndarray:
    """``negative_sample_uniform`` (data/sampling.py:7-32): ``size`` distinct non-positive items. The reference draws ranks
    among the non-positive slots with ``np.random.choice(m, size, replace=False)`` (a permutation of all m slots on the legacy
    stream — that call defines the stream and is kept) and shifts each rank past the positives below it."""
    n_free = len(choices) - len(positives)
    if n_free < size:
        raise ValueError(f'Not enough values in the range to sample "{size}" unique values.')
    slots = np.searchsorted(choices, positives)                  # positions of the positives inside `choices` (both sorted)
    ranks = np.random.choice(n_free, size=size, replace=False)
    return choices[ranks + np.searchsorted(slots - np.arange(len(slots)), ranks, side='right')]


def dataset_negative_uniform_recbole(choices: np.ndarray, size: int, positives: np.ndarray) -> np.ndarray:
    """``negative_sample_uniform_recbole`` (data/sampling.py:35-66): rejection sampling of positions in ``choices``; every round
    redraws the colliding slots with one ``np.random.randint`` call (the stream), the Python ``v in positives`` loop is a
    sorted-array membership test. As in the reference, the drawn POSITION is tested against the positive item ids."""
    n, n_pos = len(choices), len(positives)
    if n - n_pos < size:
        raise ValueError(f'Not enough values in the range to sample "{size}" unique values.')
    if (n - n_pos) * 0.5 < size:
        raise ValueError('Sampling is really inefficient either because the number of choices are small'
                         'or the number of items to sample is too high.')
    pos_sorted = np.sort(np.asarray(positives))
    neg = np.full(size, -1, dtype=np.int64)
    todo = np.arange(size)
    while todo.size:
        neg[todo] = np.random.randint(low=0, high=n, size=todo.size)
        at = np.searchsorted(pos_sorted, neg[todo])
        hit = (at < n_pos) & (pos_sorted[np.minimum(at, max(n_pos - 1, 0))] == neg[todo]) if n_pos else np.zeros(todo.size, bool)
        todo = todo[hit]
    return choices[neg]


def dataset_negative_popular(choices: np.ndarray, size: int, popularity: np.ndarray, alpha: float,
                             positives: Optional[np.ndarray] = None) -> np.ndarray:
    """``negative_sample_popular`` (data/sampling.py:69-80): with replacement, probability ~ popularity ** alpha over the
    non-positive items. ``np.random.choice(a, size, p=p)`` on the legacy stream is ``a[searchsorted(cumsum(p) / cumsum(p)[-1],
    random_sample(size), 'right')]``; written out so that the per-user work is one mask, one power and one cumsum."""
    if positives is not None:
        keep = np.ones(len(choices), dtype=bool)
        at = np.searchsorted(choices, positives)
        ok = at < len(choices)
        ok[ok] &= choices[at[ok]] == np.asarray(positives)[ok]
        keep[at[ok]] = False
        choices = choices[keep]
    p = np.power(popularity[choices], alpha)
    p = p / p.sum()
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    return choices[np.searchsorted(cdf, np.random.random_sample(size), side='right')]


def dataset_sampler_collate(user_idx: np.ndarray, pos_item_idx: np.ndarray, n_neg: int, strategy: str, choices: np.ndarray,
                            positives_of_user, popularity: Optional[np.ndarray] = None, alpha: float = 1.0):
    """One batch of ``TrainRecDataset.__getitem__`` with ``use_dataset_negative_sampler`` (data/dataset.py:379-394) + the default
    collate: rows are sampled in batch order. -> (u [B] int64, items [B, 1 + n_neg] int64, labels [B, 1 + n_neg] float64)."""
    B = len(user_idx)
    items = np.empty((B, 1 + n_neg), dtype=np.int64)
    items[:, 0] = pos_item_idx
    for r in range(B):
        pos = positives_of_user[int(user_idx[r])]
        if strategy == 'uniform':
            items[r, 1:] = dataset_negative_uniform(choices, n_neg, pos)
        elif strategy == 'uniform_recbole':
            items[r, 1:] = dataset_negative_uniform_recbole(choices, n_neg, pos)
        elif strategy == 'popular':
            items[r, 1:] = dataset_negative_popular(choices, n_neg, popularity, alpha, pos)
        else:
            raise ValueError(f'Sampling strategy "{strategy}" not yet supported.')
    labels = np.zeros((B, 1 + n_neg), dtype=np.float64)
    labels[:, 0] = 1.
    return np.asarray(user_idx, dtype=np.int64), items, labels
