"""Feature containers on the host side of the boundary.

``HostFeature`` is a minimal stand-in for the reference's ``data.Feature.Feature`` (data/Feature.py:27-295) with the same
attribute names, so that code written against either works; the engine itself only needs the duck-typed attributes
``feature_definition.type``, ``values``, ``_indices``, ``dim`` and ``n_unique_categories``.

``DeviceTable`` is what the engine keeps resident in HBM for one feature: the processed values (dense fp32 matrix, padded
int32 tag matrix, int32 category vector or CSR arrays) plus the int32 ``id -> row`` map that replaces
``np.vectorize(dict.__getitem__)`` of Feature.__getitem__ (data/Feature.py:146). Nothing is fetched from the host per batch.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import scipy.sparse as sp
import torch
from torch import nn

KINDS = ('dense', 'csr', 'categorical', 'tag')


def _type_name(t) -> str:
    return str(getattr(t, 'value', t)).lower()


class HostFeature:
    """Processed feature values + index map, mirroring data/Feature.py. ``kind`` is one of KINDS."""

    def __init__(self, name: str, kind: str, values, indices: Optional[np.ndarray] = None, n_categories: int = None,
                 unique_values: Optional[list] = None):
        """``unique_values`` (categorical): the sorted raw values whose positions are the integer categories
        (data/Feature.py:209-218) — what ``get_labels`` maps back to; evaluation groups are named after them."""
        assert kind in KINDS
        ftype = {'dense': 'vector', 'csr': 'vector', 'categorical': 'categorical', 'tag': 'tag'}[kind]
        self.feature_definition = SimpleNamespace(name=name, type=ftype)
        self.kind = kind
        self._values = values
        n = values.shape[0]
        self._n_values = n
        self._indices = np.arange(n) if indices is None else np.asarray(indices)
        if len(self._indices) != n:
            raise ValueError(f'Provided indices must match size of supplied values ({n} != {len(self._indices)})')
        if kind in ('dense', 'csr'):
            self._dim = int(values.shape[1])
        elif kind == 'tag':
            if n_categories is None:
                raise ValueError('tag features need n_categories (number of distinct tags)')
            self._dim = int(n_categories)
        else:
            self._dim = 0
        self._n_categories = n_categories if n_categories is not None else (
            int(values.max()) + 1 if kind == 'categorical' and n > 0 else None)
        self._unique_values = list(unique_values) if unique_values is not None else None
        if self._unique_values is not None and kind == 'categorical' and self._n_categories is not None \
                and len(self._unique_values) != self._n_categories:
            raise ValueError(f'{len(self._unique_values)} unique values for {self._n_categories} categories')

    @property
    def values(self):
        return self._values

    @property
    def dim(self):
        return self._dim

    @property
    def n_values(self):
        return self._n_values

    @property
    def n_unique_categories(self):
        if self.kind != 'categorical':
            raise TypeError('Only categorical features support "n_unique_categories"')
        return self._n_categories

    @property
    def unique_values(self):
        if self.kind not in ('categorical', 'tag'):
            raise TypeError('Only categorical and tag features support "unique_values"')
        if self._unique_values is None:
            return list(range(self._n_categories if self.kind == 'categorical' else self._dim))
        return self._unique_values

    def get_labels(self, values):
        """data/Feature.py:126-128: integer categories -> raw labels."""
        if self.kind != 'categorical':
            raise TypeError('Only categorical features support "get_labels"')
        uniq = self.unique_values
        return np.array([uniq[int(v)] for v in np.asarray(values).reshape(-1)])

    def __len__(self):
        return self._n_values


def feature_kind(feature) -> str:
    """Classify a reference-like Feature object (data/Feature.py:70-87 / config_classes.py:18-34)."""
    if isinstance(feature, HostFeature):
        return feature.kind
    t = _type_name(feature.feature_definition.type)
    if t == 'categorical':
        if getattr(feature, 'dim', 0) not in (0, None):
            return 'dense'           # one-hot preprocessed categorical (Feature.py:225-228)
        return 'categorical'
    if t == 'tag':
        v = np.asarray(feature.values)
        if v.ndim == 2 and v.shape[1] == feature.dim and v.max(initial=0) <= 1 and not np.issubdtype(v.dtype, np.integer):
            return 'dense'
        return 'tag'
    if sp.issparse(feature.values):
        return 'csr'
    return 'dense'


class DeviceTable(nn.Module):
    """HBM-resident view of one feature. Buffers are non-persistent: they move with ``.to(device)`` but stay out of the
    state_dict (the reference's state_dict holds parameters and BatchNorm statistics only)."""

    def __init__(self, feature):
        super().__init__()
        self.kind = feature_kind(feature)
        values = feature.values
        ids = np.asarray(getattr(feature, '_indices', np.arange(values.shape[0])))
        n = values.shape[0]
        self.n_rows = n
        # id -> row map (identity maps are dropped)
        if n == 0 or (len(ids) == n and ids[0] == 0 and np.array_equal(ids, np.arange(n))):
            rowmap = None
        else:
            rowmap = np.full(int(ids.max()) + 1, -1, dtype=np.int32)
            rowmap[ids] = np.arange(n, dtype=np.int32)
        if self.kind == 'dense':
            v = np.asarray(values)
            if v.ndim == 1:
                v = v[:, None]
            self.dim = v.shape[1]
            self.register_buffer('values', torch.from_numpy(np.ascontiguousarray(v)).float(), persistent=False)
        elif self.kind == 'csr':
            m = sp.csr_matrix(values)
            m.sort_indices()
            self.dim = m.shape[1]
            self.register_buffer('indptr', torch.from_numpy(m.indptr.astype(np.int64)), persistent=False)
            self.register_buffer('indices', torch.from_numpy(m.indices.astype(np.int32)), persistent=False)
            data = m.data.astype(np.float32)
            self.binary = bool(np.all(data == 1))
            self.register_buffer('data', None if self.binary else torch.from_numpy(data), persistent=False)
        elif self.kind == 'categorical':
            self.dim = 0
            self.n_categories = int(feature.n_unique_categories)
            cats = np.asarray(values).astype(np.int32)
            # fold category lookup into the id map: id -> category
            rowmap = cats if rowmap is None else np.where(rowmap >= 0, cats[np.clip(rowmap, 0, None)], -1).astype(np.int32)
        else:  # tag
            self.dim = int(feature.dim)
            self.pad = int(feature.dim)
            tags = np.asarray(values).astype(np.int32)
            self.T = tags.shape[1]
            self.register_buffer('tags', torch.from_numpy(np.ascontiguousarray(tags)), persistent=False)
        self.register_buffer('rowmap', None if rowmap is None else torch.from_numpy(np.ascontiguousarray(rowmap)),
                             persistent=False)
        self._transposed = None

    def n_entries(self) -> int:
        """Stored entries of a 'csr' feature / real (non-padding) tags of a 'tag' feature (cached)."""
        if getattr(self, '_n_entries', None) is None:
            self._n_entries = int(self.indices.numel()) if self.kind == 'csr' else int((self.tags != self.pad).sum())
        return self._n_entries

    def transposed(self, n_cols=None):
        """CSR form of the TRANSPOSED matrix of a 'csr' feature, on its device: (indptr int64 [dim + 1], indices int32 [nnz] = entity
        rows, data float32 [nnz] or None). Built once, on first use (the gather form of the projector's backward pass).
        A 'tag' feature: the transpose of X[entity, tag] = 1 / (tags of the entity) — the matrix an EmbeddingBag(mean) multiplies by —
        with ``n_cols`` rows (the bag's weight rows, padding row included: it has no entries)."""
        if self.kind == 'tag':
            if self._transposed is None or self._transposed[0].device != self.tags.device or self._transposed[0].numel() != n_cols + 1:
                real = self.tags != self.pad                                                  # [n_rows, T]
                cnt = real.sum(1)
                ent = torch.arange(self.n_rows, device=self.tags.device)[:, None].expand_as(self.tags)[real]
                tag = self.tags[real].long()
                val = (1.0 / cnt.float())[ent]
                order = torch.sort(tag, stable=True).indices
                t_indptr = torch.zeros(n_cols + 1, dtype=torch.int64, device=self.tags.device)
                t_indptr[1:] = torch.cumsum(torch.bincount(tag, minlength=n_cols), 0)
                self._transposed = (t_indptr, ent[order].to(torch.int32).contiguous(), val[order].contiguous())
            return self._transposed
        if self._transposed is None or self._transposed[0].device != self.indptr.device:
            counts = torch.bincount(self.indices.long(), minlength=self.dim)
            t_indptr = torch.zeros(self.dim + 1, dtype=torch.int64, device=self.indptr.device)
            t_indptr[1:] = torch.cumsum(counts, 0)
            row_of = torch.repeat_interleave(torch.arange(self.n_rows, device=self.indptr.device), self.indptr[1:] - self.indptr[:-1])
            order = torch.sort(self.indices.long(), stable=True).indices          # by column, entity rows ascending within one
            t_indices = row_of[order].to(torch.int32).contiguous()
            t_data = None if self.data is None else self.data[order].contiguous()
            self._transposed = (t_indptr, t_indices, t_data)
        return self._transposed
