"""DropoutNet (algorithms/sgd_alg.py:1617-1762; Volkovs et al., NeurIPS 2017) on the engine's kernels — SURVEY.md 8(f).4, a sibling
model of SingleBranchNet that reuses its building blocks: ``FeatureEmbedding`` content modules, ``PolyLinear`` (fp32 MFMA GEMMs with
fused bias / activation and hand-written backward), the per-slot dot and all-pairs GEMM scorers. As in the reference, the
preference input of an entity is its DENSE interaction vector (``matrix[indices].toarray()``, data/dataset.py:306-319) — produced
on the device by ``sbr_csr_rows_to_dense`` from the resident CSR — or zeros for the entities whose preferences are dropped
(``sample_training_strategy``: one draw per user / per ROW of the item index matrix from ``default_rng(sampling_seed)``).
state_dict keys as in the reference: ``{user,item}_net.pref_net.layers.linear_i.*``, ``.cont_modules.<j>.*``, ``.net.layers.linear_i.*``.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import scipy.sparse as sp
import torch
from torch import nn

from . import ops
from ._lib import call, ptr, stream
from .config import DropoutNetConfig, DropoutNetEntityConfig, DropoutNetSamplingStrategy
from .polylinear import PolyLinear
from .sbnet import FeatureEmbedding, SGDBasedRecommenderAlgorithm


class _DeviceCSR(nn.Module):
    """Interaction matrix resident in HBM (non-persistent buffers: they move with ``.to(device)``, stay out of the state_dict)."""

    def __init__(self, m):
        super().__init__()
        m = sp.csr_matrix(m)
        m.sort_indices()
        self.shape = m.shape
        self.register_buffer('indptr', torch.from_numpy(m.indptr.astype(np.int64)), persistent=False)
        self.register_buffer('indices', torch.from_numpy(m.indices.astype(np.int32)), persistent=False)
        data = m.data.astype(np.float32)
        self.register_buffer('data', None if bool(np.all(data == 1)) else torch.from_numpy(data), persistent=False)

    def dense_rows(self, ent: torch.Tensor) -> torch.Tensor:
        """ent: int64 [...] entity ids, -1 = zero vector -> float32 [..., n_cols]."""
        flat = ent.reshape(-1).long().contiguous()
        out = torch.empty(flat.numel(), self.shape[1], device=flat.device, dtype=torch.float32)
        call('sbr_csr_rows_to_dense', ptr(self.indptr), ptr(self.indices), ptr(self.data), ptr(flat), flat.numel(), self.shape[1],
             ptr(out), out.stride(0), stream())
        return out.view(*ent.shape, self.shape[1])


class DropoutNetEntity(nn.Module):
    """algorithms/sgd_alg.py:1617-1655."""

    def __init__(self, entity_name: str, preference_dim: int, features: Dict, entity_config: DropoutNetEntityConfig,
                 shared_common_dim: int):
        super().__init__()
        self.entity_name, self.entity_config, self.shared_common_dim = entity_name, entity_config, shared_common_dim
        self.pref_net = PolyLinear([preference_dim] + list(entity_config.preference_layers))
        self.pref_dim = entity_config.preference_layers[-1]
        self.cont_dim = 0
        self.cont_modules = nn.ModuleList()
        for f in entity_config.features:
            module = FeatureEmbedding.build_from_conf(f, features[f.feature_name])
            self.cont_modules.append(module)
            self.cont_dim += module.output_dim
        self._net_shape = [self.pref_dim + self.cont_dim] + list(entity_config.common_hidden_layers) + [shared_common_dim]
        self.net = PolyLinear(self._net_shape, activation_fn=entity_config.activation_fn)

    def forward(self, indices, preferences):
        pref = self.pref_net(preferences)
        cont = [m(indices).reshape(*indices.shape, -1) for m in self.cont_modules]
        return self.net(torch.cat([*cont, pref], dim=-1))


class DropoutNet(SGDBasedRecommenderAlgorithm):
    """algorithms/sgd_alg.py:1658-1762."""

    def __init__(self, config: DropoutNetConfig, dataset):
        super().__init__()
        self.config = config
        self.n_users, self.n_items = dataset.n_users, dataset.n_items
        self.user_net = DropoutNetEntity('user', preference_dim=dataset.n_items, features=dataset.user_features,
                                         entity_config=config.user, shared_common_dim=config.shared_common_dim)
        self.item_net = DropoutNetEntity('item', preference_dim=dataset.n_users, features=dataset.item_features,
                                         entity_config=config.item, shared_common_dim=config.shared_common_dim)
        # get_user_interaction_vectors / get_item_interaction_vectors of the dataset (data/dataset.py:260-273)
        self._user_rows = _DeviceCSR(dataset.user_sampling_matrix_train)
        self._item_rows = _DeviceCSR(dataset.item_sampling_matrix_train)
        self._rng = np.random.default_rng(config.sampling_seed)
        self.name = 'DropoutNet'

    def sample_training_strategy(self, n_samples):
        if self.training:
            return self._rng.choice(DropoutNetSamplingStrategy.list(), size=n_samples, replace=True)
        return np.full(n_samples, fill_value=DropoutNetSamplingStrategy.Normal.value)      # validation: all information

    def _preferences(self, idxs: torch.Tensor, rows: _DeviceCSR, strategy=None) -> torch.Tensor:
        """Dense preference vectors of ``idxs`` ([B] users or [B, N] items); entities (rows of ``idxs``) drawn NoPreference get
        zeros. One strategy per leading row, as in the reference (``len(idxs)`` draws)."""
        if not idxs.is_cuda:
            raise RuntimeError('DropoutNet (HIP engine) needs CUDA(HIP) index tensors')
        if strategy is None:
            strategy = self.sample_training_strategy(len(idxs))
        keep = torch.from_numpy(np.asarray(strategy) == DropoutNetSamplingStrategy.Normal.value).to(idxs.device)
        ent = torch.where(keep.view(-1, *([1] * (idxs.ndim - 1))), idxs.long(), torch.full_like(idxs.long(), -1))
        return rows.dense_rows(ent)

    def get_user_representations(self, u_idxs, strategy=None):
        return self.user_net(u_idxs, self._preferences(u_idxs, self._user_rows, strategy))

    def get_item_representations(self, i_idxs, strategy=None):
        return self.item_net(i_idxs, self._preferences(i_idxs, self._item_rows, strategy))

    def combine_user_item_representations(self, u_repr, i_repr):
        return (ops.ScoreAllFn if i_repr.ndim == 2 else ops.ScoreDotFn).apply(u_repr, i_repr)

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return DropoutNet(DropoutNetConfig.from_dict(conf), dataset)
