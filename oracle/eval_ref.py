"""Oracle (test infrastructure): CPU restatement of full-catalogue scoring and the metric definitions.

Reference files followed (relative to /root/reference):
  eval/eval.py:203-222       -> ``score_users`` (item reprs once, per user batch: scores, -inf mask)
  eval/eval.py:320           -> ``topk``        (torch.topk, sorted, largest)
  eval/metrics.py:4-36       -> ``recall_at_k``
  eval/metrics.py:39-67      -> ``precision_at_k``
  eval/metrics.py:70-105     -> ``ndcg_at_k``

The live reference computes its metrics in the third-party package ``rmet`` (eval/eval.py:99-102), an
unpinned git dependency that is absent offline: with respect to ``rmet`` the metric arithmetic below is
PARITY UNPINNED; it follows the in-repo definition eval/metrics.py instead (binary relevance).
"""
from __future__ import annotations

import numpy as np
import torch


def masked_scores(u_repr: torch.Tensor, i_repr: torch.Tensor, exclude_dense: np.ndarray) -> torch.Tensor:
    """eval/eval.py:217-220: out = u @ i^T ; out[mask] = -inf."""
    out = u_repr @ i_repr.t()
    out = out.clone()
    out[torch.from_numpy(np.asarray(exclude_dense, dtype=bool))] = -torch.inf
    return out


def topk(scores: torch.Tensor, k: int):
    t = torch.topk(scores, k, largest=True, sorted=True)
    return t.values, t.indices


def _hits(y_true: torch.Tensor, idx_topk: torch.Tensor) -> torch.Tensor:
    rows = torch.arange(y_true.shape[0]).unsqueeze(-1)
    return y_true[rows, idx_topk]


def recall_at_k(y_true: torch.Tensor, idx_topk: torch.Tensor) -> torch.Tensor:
    """eval/metrics.py:22-27: hits@k / n_positives, NaN (no positives) -> 0."""
    r = _hits(y_true, idx_topk).sum(-1) / y_true.sum(-1)
    r[torch.isnan(r)] = 0.
    return r


def precision_at_k(y_true: torch.Tensor, idx_topk: torch.Tensor) -> torch.Tensor:
    """eval/metrics.py:57-59: hits@k / k."""
    return _hits(y_true, idx_topk).sum(-1) / idx_topk.shape[-1]


def ndcg_at_k(y_true: torch.Tensor, idx_topk: torch.Tensor) -> torch.Tensor:
    """eval/metrics.py:88-98: DCG = sum_r y[top_r] / log2(r + 2); IDCG = the same over the k largest labels;
    NaN -> 0; clamp to <= 1."""
    k = idx_topk.shape[-1]
    disc = 1. / torch.log2(torch.arange(2, k + 2).float())
    dcg = (_hits(y_true, idx_topk) * disc).sum(-1)
    idcg = (y_true.topk(k).values * disc).sum(-1)
    n = dcg / idcg
    n[torch.isnan(n)] = 0.
    return n.clamp(max=1.)


def evaluate(u_repr: torch.Tensor, i_repr: torch.Tensor, exclude_dense: np.ndarray, labels_dense: np.ndarray,
             ks=(1, 10, 20)):
    """One user batch of eval/eval.py:212-222 with the metric definitions above. Returns per-user arrays."""
    s = masked_scores(u_repr, i_repr, exclude_dense)
    y = torch.from_numpy(np.asarray(labels_dense, dtype=np.float32))
    kmax = min(max(ks), s.shape[1])
    vals, idx = topk(s, kmax)
    out = {'topk_scores': vals, 'topk_idx': idx}
    for k in ks:
        kk = min(k, kmax)
        out[f'ndcg@{k}'] = ndcg_at_k(y, idx[:, :kk])
        out[f'recall@{k}'] = recall_at_k(y, idx[:, :kk])
        out[f'precision@{k}'] = precision_at_k(y, idx[:, :kk])
    return out


def group_metrics(per_user: dict, group_name: str, labels) -> dict:
    """eval/eval.py:106-119 (``_calculate_group_metrics``): the per-user metric arrays restricted to the users of each
    label of one categorical user feature; string labels are lower-cased (:112-113). ``per_user``: {'ndcg@10': array[Bu], ...}
    as ``evaluate`` returns them; ``labels``: one label per user of the batch, in batch order. Keys follow rmet's flattening
    '{group}_{label}/{metric}' (rmet is absent offline: the key layout is PARITY UNPINNED; the values are the same metric
    functions applied to a row subset)."""
    labels = np.array([lbl.lower() if isinstance(lbl, str) else lbl for lbl in labels])
    out = {}
    for lbl in sorted(set(labels.tolist())):
        sel = np.flatnonzero(labels == lbl)
        for k, v in per_user.items():
            if '@' in k and not k.startswith('topk'):
                out[f'{group_name}_{lbl}/{k}'] = np.asarray(v)[sel]
    return out


def natural_sorted(keys):
    """``natsorted(keys)`` of eval/eval.py:160 restated (natsort's default algorithm: runs of digits compare as integers)."""
    import re
    return sorted(keys, key=lambda s: [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', s)])
