"""Oracle (test infrastructure): CPU restatement of the recommendation losses.

Reference files followed (relative to /root/reference):
  train/rec_losses.py:40-58    -> ``bce_loss``
  train/rec_losses.py:61-83    -> ``bpr_loss``
  train/rec_losses.py:86-113   -> ``sampled_softmax_loss``
  train/rec_losses.py:12-37    -> ``RefRecLoss``
InfoNCE (train/regularization_losses.py:14-43) lives in ``model_ref.info_nce``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _reduce(x: torch.Tensor, aggregator: str) -> torch.Tensor:
    return x.mean() if aggregator == 'mean' else x.sum()


def bce_loss(logits: torch.Tensor, labels: torch.Tensor, aggregator: str = 'mean') -> torch.Tensor:
    """rec_losses.py:56 — BCE-with-logits over all B*N slots. Labels are float64 (dataloader.py:196), so the
    per-element loss  max(x,0) - x*y + log1p(exp(-|x|))  is promoted to float64 before the reduction."""
    x = logits.flatten()
    y = labels.flatten()
    per = torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))
    return _reduce(per, aggregator)


def bpr_loss(logits: torch.Tensor, labels: torch.Tensor, aggregator: str = 'mean') -> torch.Tensor:
    """rec_losses.py:73-83 — softplus(-(pos - neg)) over the B*n_neg differences with the (float64) label of
    the positive column repeated as target (== 1)."""
    diff = (logits[:, :1] - logits[:, 1:]).flatten()
    target = torch.repeat_interleave(labels[:, 0], logits.shape[1] - 1)
    per = torch.clamp(diff, min=0) - diff * target + torch.log1p(torch.exp(-diff.abs()))
    return _reduce(per, aggregator)


def sampled_softmax_loss(logits: torch.Tensor, labels: torch.Tensor, aggregator: str = 'mean',
                         n_items: int = None, neg_train: int = 4, strategy: str = 'uniform') -> torch.Tensor:
    """rec_losses.py:101-113 — -x_pos + logsumexp(x); with strategy 'uniform' the negatives are shifted by
    log(n_items / n_neg) first (the reference does that IN PLACE on ``logits``; this restatement is
    out-of-place, the value and gradient are the same)."""
    x = logits
    if strategy == 'uniform':
        shift = torch.zeros_like(x)
        shift[:, 1:] = math.log(n_items / neg_train)
        x = x + shift
    per = -x[:, 0] + torch.logsumexp(x, dim=-1)
    return _reduce(per, aggregator)


class RefRecLoss:
    """RecommenderSystemLoss + enum (rec_losses.py:12-37, 116-119)."""

    def __init__(self, kind: str, n_items: int = None, aggregator: str = 'mean',
                 train_neg_strategy: str = 'uniform', neg_train: int = 4):
        assert aggregator in ('mean', 'sum'), 'Type of Aggregator not yet defined'
        assert train_neg_strategy in ('uniform', 'uniform_recbole'), 'Type of Negative Strategy not currently supported'
        assert kind in ('bce', 'bpr', 'sampled_softmax')
        self.kind, self.n_items, self.aggregator = kind, n_items, aggregator
        self.train_neg_strategy, self.neg_train = train_neg_strategy, neg_train

    def compute_loss(self, logits, labels):
        if self.kind == 'bce':
            return bce_loss(logits, labels, self.aggregator)
        if self.kind == 'bpr':
            return bpr_loss(logits, labels, self.aggregator)
        return sampled_softmax_loss(logits, labels, self.aggregator, self.n_items, self.neg_train,
                                    self.train_neg_strategy)
