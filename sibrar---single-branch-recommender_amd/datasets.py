"""Dataset views and loaders on the host side of the hot path.

* ``SyntheticDataset`` generates the synthetic workloads of SURVEY.md §8(d) / BASELINE.json (no dataset can be fetched
  offline) and exposes exactly the attributes the plugin reads from the reference's datasets
  (algorithms/sgd_alg.py:2021-2067, eval/eval.py:209,219, data/dataset.py:325-453).
* ``NegativeSamplingDataLoader`` yields the same ``(u_idxs i64 [B], i_idxs i64 [B, 1+n_neg], labels f64)`` batches as the
  reference's default loader (data/dataloader.py:134-198 with ``shuffle=True``), bit-exactly for the same seeds, using the
  vectorised collate of sampling.py.
"""
from __future__ import annotations

import atexit
import os
import weakref
from types import SimpleNamespace
from typing import Dict, Optional, Sequence

import numpy as np
import scipy.sparse as sp
import torch

from .features import HostFeature
from .sampling import (DevicePositiveIndex, PositiveIndex, dataset_sampler_collate, is_arange, loader_epoch_order,
                       recbole_negative_collate, uniform_negative_collate)


def synthetic_interactions(n_users: int, n_items: int, nnz: int, seed: int = 0, item_popularity: float = 0.0) -> sp.csr_matrix:
    """Per-user degree from a log-normal scaled to the target density; items uniform, or — ``item_popularity`` = a > 0 — drawn with
    probability proportional to 1 / rank^a over a random ranking of the items ("Zipf 1.0 for popularity realism":
    something a model can learn, so that a trained model's NDCG is above the noise floor)."""
    rng = np.random.default_rng(seed)
    deg = rng.lognormal(mean=0., sigma=1., size=n_users)
    deg = np.maximum(1, np.round(deg / deg.sum() * nnz)).astype(np.int64)
    deg = np.minimum(deg, max(1, n_items // 2))
    rows = np.repeat(np.arange(n_users, dtype=np.int64), deg)
    if item_popularity > 0:
        p = 1.0 / np.arange(1, n_items + 1, dtype=np.float64) ** float(item_popularity)
        ranking = np.random.default_rng(seed + 11).permutation(n_items)
        cols = ranking[np.searchsorted(np.cumsum(p / p.sum()), rng.random(rows.size), side='right').clip(max=n_items - 1)]
    else:
        cols = rng.integers(0, n_items, size=rows.size)
    m = sp.csr_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)), shape=(n_users, n_items))
    m.sum_duplicates()
    m.data[:] = 1
    m.sort_indices()
    return m


class SyntheticDataset:
    def __init__(self, n_users: int, n_items: int, nnz: int, item_dense: Dict[str, int] = None,
                 item_tags: Dict[str, tuple] = None, user_categorical: Dict[str, int] = None, seed: int = 0,
                 n_negative_samples: int = 10, negative_sampling_strategy: str = 'uniform_recbole',
                 holdout_per_user: int = 0, item_popularity: float = 0.0):
        self.n_users, self.n_items = n_users, n_items
        inter = synthetic_interactions(n_users, n_items, nnz, seed, item_popularity)
        self.holdout = None
        if holdout_per_user > 0:
            # last `holdout_per_user` items of each user (>= 2 interactions) become the evaluation labels
            rng = np.random.default_rng(seed + 7)
            lil_rows, lil_cols = [], []
            keep = inter.copy().tolil()
            for u in range(n_users):
                cols = inter.indices[inter.indptr[u]:inter.indptr[u + 1]]
                if len(cols) > holdout_per_user:
                    h = rng.choice(cols, size=holdout_per_user, replace=False)
                    lil_rows += [u] * len(h)
                    lil_cols += h.tolist()
            self.holdout = sp.csr_matrix((np.ones(len(lil_rows), dtype=np.int8), (lil_rows, lil_cols)), shape=inter.shape)
            inter = (inter - self.holdout).tocsr()
            inter.eliminate_zeros()
        self.interaction_matrix = inter.tocoo()
        self.user_sampling_matrix = inter
        self.user_sampling_matrix_train = inter
        self.item_sampling_matrix_train = sp.csr_matrix(inter.T)
        self.items_in_split = np.arange(n_items)
        self.users_in_split = np.arange(n_users)
        self.n_items_in_split, self.n_users_in_split = n_items, n_users
        self.is_cold_start_user = self.is_cold_start_item = False
        self.n_negative_samples = n_negative_samples
        self.negative_sampling_strategy = negative_sampling_strategy
        rng = np.random.default_rng(seed + 1)
        self.item_features, self.user_features = {}, {}
        for name, dim in (item_dense or {}).items():
            self.item_features[name] = HostFeature(name, 'dense', rng.standard_normal((n_items, dim), dtype=np.float32))
        rng = np.random.default_rng(seed + 2)
        for name, (n_tags, max_tags) in (item_tags or {}).items():
            cnt = rng.integers(1, max_tags + 1, size=n_items)
            tags = np.full((n_items, max_tags), n_tags, dtype=np.int64)
            for j in range(max_tags):
                draw = rng.integers(0, n_tags, size=n_items)
                tags[:, j] = np.where(j < cnt, draw, n_tags)
            self.item_features[name] = HostFeature(name, 'tag', tags, n_categories=n_tags)
        for name, n_cat in (user_categorical or {}).items():
            self.user_features[name] = HostFeature(name, 'categorical', rng.integers(0, n_cat, size=n_users), n_categories=n_cat)

    def __len__(self):
        return self.interaction_matrix.nnz

    def eval_view(self):
        """FullEvalDataset-like view (data/dataset.py:399-453): labels = held-out interactions, exclusions = train."""
        assert self.holdout is not None, 'build the dataset with holdout_per_user > 0'
        return SimpleNamespace(n_users=self.n_users, n_items=self.n_items, items_in_split=self.items_in_split,
                               users_in_split=self.users_in_split, n_items_in_split=self.n_items,
                               n_users_in_split=self.n_users, user_sampling_matrix=self.holdout,
                               exclude_data=self.user_sampling_matrix_train.astype(bool),
                               user_features=self.user_features, item_features=self.item_features)


_LIVE_LOADERS = weakref.WeakSet()


@atexit.register
def _close_loaders():
    for ld in list(_LIVE_LOADERS):
        ld.close()


class NegativeSamplingDataLoader:
    """Default training loader of the reference (data/dataloader.py:134-198 + shuffle) with the vectorised collate."""

    def __init__(self, dataset, batch_size: int = 256, shuffle: bool = True, strategy: Optional[str] = None,
                 rank: int = 0, world: int = 1, max_batches: Optional[int] = None, device=None, prefetch: int = 0,
                 draw_fn=None, prepare_fn=None, dp_sampling: str = 'global', use_dataset_negative_sampler: bool = False):
        """``device``: run the collision test of the collate on that GPU (DevicePositiveIndex) instead of numpy.
        ``prefetch`` > 0: a producer thread prepares up to that many batches ahead (single producer, so the RNG streams are
        consumed in the same order as without it).
        ``draw_fn(u_shape, i_shape)``: optional callable (engine.FusedTrainStep.draw) evaluated by the producer for every
        batch; its result is yielded as a 4th element so that the modality draw also leaves the launch thread.
        ``dp_sampling`` (world > 1): 'global' — ``batch_size`` is the GLOBAL batch; every rank draws the whole global batch
        from the same random streams and keeps rows rank::world, so the union over ranks is bit-identical to the 1-GPU
        batch (host cost grows with world). 'local' — ``batch_size`` is the PER-RANK batch; a rank collates only its own
        contiguous slice of the shared epoch order and draws negatives from its own global-numpy stream (seed it per rank):
        host cost independent of world, the weak-scaling mode of bench.py.
        ``prepare_fn(u, i, labels)``: same, but with the batch itself (engine.FusedTrainStep.prepare: draw + launch plan +
        uploads on a side stream) — the 4th element is then a ready ``PreparedBatch``."""
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.strategy = strategy or dataset.negative_sampling_strategy
        # ``use_dataset_negative_sampler``: the negatives come from the dataset-level samplers, one call per interaction in
        # batch order (TrainRecDataset.__getitem__, data/dataset.py:379-394; strategies uniform / uniform_recbole / popular)
        self.dataset_sampler = bool(use_dataset_negative_sampler)
        allowed = ('uniform', 'uniform_recbole', 'popular') if self.dataset_sampler else ('uniform_recbole', 'uniform')
        if self.strategy not in allowed:
            raise ValueError(f'sampling strategy {self.strategy} not supported for dataloader sampling!')
        self.n_neg = dataset.n_negative_samples
        coo = dataset.interaction_matrix
        self.rows, self.cols = coo.row.astype(np.int64), coo.col.astype(np.int64)
        self.positives = (DevicePositiveIndex(dataset.user_sampling_matrix, device) if device is not None
                          else PositiveIndex(dataset.user_sampling_matrix))
        self.rank, self.world, self.max_batches = rank, world, max_batches
        self._device = None
        if device is not None:
            d = torch.device(device)
            self._device = torch.device('cuda', d.index if d.index is not None else torch.cuda.current_device())
        if dp_sampling not in ('global', 'local'):
            raise ValueError(f'dp_sampling {dp_sampling!r}')
        self.dp_sampling = dp_sampling
        self.prefetch = prefetch
        self.draw_fn = draw_fn
        self.prepare_fn = prepare_fn
        self._prepare_takes_key = None
        self._identity_items = is_arange(np.asarray(dataset.items_in_split))
        self._pos_rows, self._pop = None, None
        self._native = None          # native_loader.NativeBatchProducer, created on first use

    def __len__(self):
        if self.dp_sampling == 'local' and self.world > 1:
            n = len(self.rows) // (self.batch_size * self.world)      # whole global steps only: every rank runs the same count
        else:
            n = (len(self.rows) + self.batch_size - 1) // self.batch_size
            if self.world > 1 and 0 < len(self.rows) % self.batch_size < self.world:
                n -= 1               # 'global' sampling: a last global batch with fewer rows than ranks is dropped (see _produce)
        return n if self.max_batches is None else min(n, self.max_batches)

    def _native_producer(self):
        """The native batch producer (native_loader.py / csrc/producer.hip) when ``prepare_fn`` is a FusedTrainStep's ``prepare``
        and the loader runs its default path; None otherwise (the Python pipeline below serves every other configuration)."""
        fused = getattr(self.prepare_fn, '__self__', None)
        if fused is None or type(fused).__name__ != 'FusedTrainStep' or self.draw_fn is not None:
            return None
        from . import native_loader
        if not native_loader.eligible(self, fused):
            return None
        if self._native is None or self._native.fused is not fused or self._native.B != self.batch_size:
            if self._native is not None:
                self._native.close()
            self._native = native_loader.NativeBatchProducer(self, fused)
        return self._native

    def _native_iter(self, prod):
        n = len(self.rows)
        order = loader_epoch_order(n) if self.shuffle else np.arange(n)
        rows_e, cols_e = (self.rows[order], self.cols[order]) if self.shuffle else (self.rows, self.cols)
        local = self.dp_sampling == 'local' and self.world > 1
        first, stride = (self.rank * self.batch_size, self.world * self.batch_size) if local else (0, self.batch_size)
        prod.start(rows_e, cols_e, first, stride, len(self))
        _LIVE_LOADERS.add(self)
        try:
            while True:
                pb = prod.next_batch()
                if pb is None:
                    break
                yield None, None, None, pb
        finally:
            prod.stop()

    def __iter__(self):
        if self.prefetch > 0 and self.prepare_fn is not None:
            prod = self._native_producer()
            if prod is not None:
                self.close()
                return self._native_iter(prod)
        if self.prefetch <= 0:
            return self._produce()
        import queue
        import threading
        self.close()                                   # one live pipeline per loader
        done = object()
        stop = threading.Event()

        def put(q, item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def stage(source, sink, fn):
            try:
                # the HIP current device is per thread (a new thread starts on device 0): kernels and copies of this thread
                # must be issued with the loader's device current — rank r of a multi-GPU job drives cuda:r
                if self._device is not None and torch.cuda.is_available():
                    torch.cuda.set_device(self._device)
                for b in source:
                    if stop.is_set() or not put(sink, fn(b)):
                        return
                put(sink, done)
            except BaseException as e:      # surface producer errors in the consumer
                put(sink, e)

        def drain(q):
            while not stop.is_set():
                try:
                    b = q.get(timeout=0.05)
                except queue.Empty:
                    continue
                if b is done:
                    return
                if isinstance(b, BaseException):
                    raise b
                yield b

        out = queue.Queue(maxsize=self.prefetch)
        threads = []
        two_stage = self.batch_size * (1 + self.n_neg) >= 32768
        if self.prepare_fn is None or not two_stage:
            # small batches: every numpy call is too short to release the GIL, a second producer thread only adds hand-offs
            threads.append(threading.Thread(target=stage, args=(self._produce(), out, lambda b: b), daemon=True))
        else:
            # two pipeline stages, one thread each: collate (global numpy stream) -> prepare (entity streams, uploads). Each
            # random stream is still consumed by exactly one thread in batch order.
            mid = queue.Queue(maxsize=self.prefetch)
            threads.append(threading.Thread(target=stage, args=(self._produce(prepare=False), mid, lambda b: b), daemon=True))
            threads.append(threading.Thread(target=stage, args=(drain(mid), out, lambda b: (*b, self._prepare(*b))),
                                            daemon=True))
        for th in threads:
            th.start()
        self._live = (stop, threads)
        _LIVE_LOADERS.add(self)
        return drain(out)

    def _prepare(self, u, i, l):
        # the collates of this loader always produce "first column positive, the rest negative" labels: tell prepare()
        # (engine.FusedTrainStep.prepare keeps one device copy per shape) when it understands the promise
        if self._prepare_takes_key is None:
            import inspect
            try:
                self._prepare_takes_key = 'labels_key' in inspect.signature(self.prepare_fn).parameters
            except (TypeError, ValueError):
                self._prepare_takes_key = False
        if self._prepare_takes_key:
            return self.prepare_fn(u, i, l, labels_key='first_column_positive')
        return self.prepare_fn(u, i, l)

    def close(self):
        """Stop the producer threads of the current iteration (also run at interpreter exit: a thread that is inside a
        HIP call while Python finalises aborts the process)."""
        if getattr(self, '_native', None) is not None:
            self._native.stop()
        live = getattr(self, '_live', None)
        if live is None:
            return
        stop, threads = live
        stop.set()
        for th in threads:
            th.join(timeout=5.0)
        self._live = None

    def _produce(self, prepare: bool = True):
        n = len(self.rows)
        order = loader_epoch_order(n) if self.shuffle else np.arange(n)
        # the epoch's (user, item) pairs in visiting order, gathered ONCE: per batch a contiguous slice instead of two random
        # gathers over the whole interaction list
        rows_e, cols_e = (self.rows[order], self.cols[order]) if self.shuffle else (self.rows, self.cols)
        local = self.dp_sampling == 'local' and self.world > 1
        for b in range(len(self)):
            if local:       # rank r owns the r-th contiguous chunk of every global step (an incomplete last global step is dropped)
                lo = (b * self.world + self.rank) * self.batch_size
            else:
                lo = b * self.batch_size
            bu, bi = rows_e[lo:lo + self.batch_size], cols_e[lo:lo + self.batch_size]
            if self.dataset_sampler:
                if self._pos_rows is None:
                    m = self.dataset.user_sampling_matrix
                    self._pos_rows = [m.indices[m.indptr[r]:m.indptr[r + 1]] for r in range(m.shape[0])]
                    if self.strategy == 'popular':
                        pop = np.asarray(m.sum(axis=0)).flatten()
                        self._pop = getattr(self.dataset, 'pop_distribution', pop / pop.sum())
                u, i, l = dataset_sampler_collate(bu, bi, self.n_neg, self.strategy, np.asarray(self.dataset.items_in_split),
                                                  self._pos_rows, self._pop,
                                                  float(getattr(self.dataset, 'sampling_popularity_squashing_factor', 1.0)))
            elif self.strategy == 'uniform_recbole':
                u, i, l = recbole_negative_collate(bu, bi, self.n_neg, self.dataset.items_in_split, self.positives,
                                                   self._identity_items)
            else:
                u, i, l = uniform_negative_collate(bu, bi, self.n_neg, self.dataset.n_items, self.positives)
            # data parallel: every rank consumes the same global streams and keeps its slice (parallel.shard_batch)
            if self.world > 1 and not local:
                # every rank must hold the SAME number of rows: the step divides each rank's loss by its own row count and
                # averages the gradients over the ranks, which is the gradient of the global-batch mean only for equal shards —
                # and a rank without rows would leave the others waiting in the all-reduce. The (at most world - 1) surplus
                # rows of an incomplete last global batch are dropped on every rank alike.
                keep = len(u) // self.world * self.world
                if keep == 0:
                    continue
                u, i, l = u[:keep][self.rank::self.world], i[:keep][self.rank::self.world], l[:keep][self.rank::self.world]
            if not prepare:
                yield torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(l)
            elif self.prepare_fn is not None:
                tu, ti, tl = torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(l)
                yield tu, ti, tl, self._prepare(tu, ti, tl)
            elif self.draw_fn is not None:
                yield torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(l), self.draw_fn(u.shape, i.shape)
            else:
                yield torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(l)
