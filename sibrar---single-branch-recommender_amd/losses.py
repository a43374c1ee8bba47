"""Recommendation losses and InfoNCE with the reference's class names and semantics, computed by the fused HIP kernels
(forward value + gradient w.r.t. the logits in one kernel each).

Mirrors train/rec_losses.py:12-119 and train/regularization_losses.py:8-51.
"""
from __future__ import annotations

import math
from abc import ABC, abstractmethod
from enum import Enum

import torch
from torch import nn

from . import ops


class RecommenderSystemLoss(ABC):
    def __init__(self, n_items: int = None, aggregator: str = 'mean', train_neg_strategy: str = 'uniform',
                 neg_train: int = 4):
        assert aggregator in ['mean', 'sum'], "Type of Aggregator not yet defined"
        assert train_neg_strategy in ['uniform', 'uniform_recbole'], "Type of Negative Strategy not currently supported"
        super().__init__()
        self.n_items = n_items
        self.aggregator = aggregator
        self.train_neg_strategy = train_neg_strategy
        self.neg_train = neg_train

    @abstractmethod
    def compute_loss(self, logits: torch.Tensor, labels: torch.Tensor):
        pass

    @staticmethod
    def build_from_conf(conf, dataset):
        """rec_losses.py:27-37. ``conf`` needs ``learn.rec_loss``, ``learn.loss_aggregator`` and the dataset section's
        ``negative_sampling_strategy`` / ``n_negative_samples`` (dict or attribute access)."""
        def get(o, k, default=None):
            return o.get(k, default) if isinstance(o, dict) else getattr(o, k, default)
        learn, ds = get(conf, 'learn'), get(conf, 'dataset')
        cls = RecommenderSystemLossesEnum[get(learn, 'rec_loss')].value
        return cls(n_items=dataset.n_items, aggregator=get(learn, 'loss_aggregator', 'mean'),
                   train_neg_strategy=get(ds, 'negative_sampling_strategy', 'uniform'),
                   neg_train=get(ds, 'n_negative_samples', 4))


class RecBinaryCrossEntropy(RecommenderSystemLoss):
    def compute_loss(self, logits, labels):
        B, N = logits.shape
        scale = 1.0 / (B * N) if self.aggregator == 'mean' else 1.0
        return ops.RecLossFn.apply(logits, labels, ops.LOSS_CODES['bce'], scale, 0.0)


class RecBayesianPersonalizedRankingLoss(RecommenderSystemLoss):
    def compute_loss(self, logits, labels):
        B, N = logits.shape
        scale = 1.0 / (B * (N - 1)) if self.aggregator == 'mean' else 1.0
        return ops.RecLossFn.apply(logits, labels, ops.LOSS_CODES['bpr'], scale, 0.0)


class RecSampledSoftmaxLoss(RecommenderSystemLoss):
    def compute_loss(self, logits, labels):
        B, N = logits.shape
        scale = 1.0 / B if self.aggregator == 'mean' else 1.0
        # rec_losses.py:104-105 (the reference applies the shift in place on ``logits``; here it is applied inside the kernel)
        shift = math.log(self.n_items / self.neg_train) if self.train_neg_strategy == 'uniform' else 0.0
        return ops.RecLossFn.apply(logits, labels, ops.LOSS_CODES['sampled_softmax'], scale, shift)


class RecommenderSystemLossesEnum(Enum):
    bce = RecBinaryCrossEntropy
    bpr = RecBayesianPersonalizedRankingLoss
    sampled_softmax = RecSampledSoftmaxLoss


class InfoNCE(nn.Module):
    """train/regularization_losses.py:8-43 for inputs [..., N, D] (groups = product of the leading dims)."""

    def __init__(self, temperature: float = 1., loss_aggregator: str = 'mean'):
        super().__init__()
        self.temperature = temperature
        self.loss_aggregator = loss_aggregator

    def forward(self, first_emb, second_emb):
        N, D = first_emb.shape[-2], first_emb.shape[-1]
        G = first_emb.numel() // (N * D)
        e = torch.stack([first_emb.reshape(-1, D), second_emb.reshape(-1, D)], dim=1).contiguous()
        return ops.InfoNCEFn.apply(e, float(self.temperature), self.loss_aggregator == 'mean', G, N)


class ZeroLossModule(nn.Module):
    def forward(self, *args, **kwargs):
        return torch.tensor([0])
