"""Full-catalogue evaluation with the reference's hooks: ``FullEvaluator`` and ``evaluate_recommender_algorithm``
(eval/eval.py:20-168, 171-227).

The reference scores a user batch against all items (``einsum('be,ce->bc')``), sets ``out[exclude_data[u]] = -inf`` from a
host-densified CSR mask, and hands the dense [Bu, I_s] logits plus dense label rows to the third-party ``rmet.calculate``.
Here the exclusion CSR and the label CSR are resident on the device; per user batch the engine runs either
  * ``scorer='fp32'``      : fp32-MFMA GEMM -> CSR mask kernel -> exact radix-select top-k, or
  * ``scorer='fp16_fused'``: the fused fp16-MFMA score+mask+top-k kernel (scores never written),
followed by the ranking-metric kernel (NDCG / recall / precision as defined in eval/metrics.py:4-105; ``rmet`` itself is
absent offline, so w.r.t. ``rmet`` the metric arithmetic is parity-unpinned). ``eval_batch`` keeps the reference's
dense-logits entry point for callers that already hold a score matrix.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Optional, Sequence

import numpy as np
import scipy.sparse as sp
import torch

from . import ops

SUPPORTED_METRICS = ('ndcg', 'precision', 'recall', 'f_score', 'hitrate', 'coverage')


class _Cfg:
    def __init__(self, top_k=(1, 3, 5, 10, 20, 50, 100), metrics=SUPPORTED_METRICS, calculate_std=True):
        self.top_k, self.metrics, self.calculate_std = list(top_k), list(metrics), calculate_std


def _csr_to_device(m: sp.spmatrix, device):
    m = sp.csr_matrix(m)
    m.sort_indices()
    m.eliminate_zeros()
    return (torch.from_numpy(m.indptr.astype(np.int64)).to(device), torch.from_numpy(m.indices.astype(np.int32)).to(device))


class FullEvaluator:
    """eval/eval.py:20-168 — accumulates per-user metric arrays over batches; ``get_results`` averages them."""

    def __init__(self, config=None, evaluator_name: str = None, dataset=None):
        self.config = config if config is not None else _Cfg()
        self.name = evaluator_name
        self.dataset = dataset
        invalid = set(self.config.metrics) - set(SUPPORTED_METRICS)
        if invalid:
            raise ValueError(f'Metric(s) {invalid} are not supported. Select metrics from {SUPPORTED_METRICS}.')
        self._ks = sorted(set(int(k) for k in self.config.top_k))
        self._labels_dev = None
        self._reset()

    def _reset(self):
        self._results = defaultdict(list)
        self._topk = []

    def _key(self, metric, k):
        base = f'{metric}@{k}'
        return f'{self.name}/{base}' if self.name else base

    def _labels(self, device):
        if self._labels_dev is None or self._labels_dev[0].device != torch.device(device):
            ds = self.dataset
            cached = getattr(ds, '_labels_dev', None)          # evaluators come and go (one per evaluation), the split does not
            if cached is None or cached[0].device != torch.device(device):
                lab = sp.csr_matrix(ds.user_sampling_matrix)[:, np.asarray(ds.items_in_split)]
                cached = _csr_to_device(lab, device)
                try:
                    ds._labels_dev = cached
                except Exception:
                    pass
            self._labels_dev = cached
        return self._labels_dev

    def eval_topk(self, u_idxs: torch.Tensor, topk_idx: torch.Tensor):
        """Engine entry point: per-user top-k item positions (int32 [Bu, kmax], kmax >= max(top_k))."""
        kmax = topk_idx.shape[1]
        ks = [k for k in self._ks if k <= kmax]
        indptr, indices = self._labels(topk_idx.device)
        m = ops.rank_metrics(topk_idx.contiguous(), u_idxs.long().contiguous(), indptr, indices, ks)   # [3, n_ks, Bu]
        for qi, k in enumerate(ks):
            nd, rc, pr = m[0, qi], m[1, qi], m[2, qi]
            if 'ndcg' in self.config.metrics:
                self._results[self._key('ndcg', k)].append(nd)
            if 'recall' in self.config.metrics:
                self._results[self._key('recall', k)].append(rc)
            if 'precision' in self.config.metrics:
                self._results[self._key('precision', k)].append(pr)
            if 'hitrate' in self.config.metrics:
                self._results[self._key('hitrate', k)].append((pr > 0).float())
            if 'f_score' in self.config.metrics:
                den = pr + rc
                self._results[self._key('f_score', k)].append(torch.where(den > 0, 2 * pr * rc / den.clamp_min(1e-30), den))
        if 'coverage' in self.config.metrics:
            self._topk.append(topk_idx)

    def eval_batch(self, u_idxs: torch.Tensor, logits: torch.Tensor, y_true: torch.Tensor = None):
        """Reference entry point (eval.py:121-138): dense [Bu, I_s] logits (already masked). ``y_true`` is ignored when the
        evaluator was built with a dataset (labels come from the resident CSR); without a dataset the dense rows are used."""
        if y_true is not None and logits.shape != y_true.shape:
            raise AttributeError(f'logits and true labels must have the same shape ({logits.shape} != {y_true.shape})')
        if len(u_idxs) != len(logits):
            raise AttributeError('assumed batch size is not equal for user indices, logits and true labels')
        kmax = min(max(self._ks), logits.shape[1])
        _, idx = ops.topk_rows(logits.float().contiguous(), kmax)
        if self.dataset is None:
            lab = sp.csr_matrix(y_true.detach().cpu().numpy() > 0)
            self._labels_dev = _csr_to_device(lab, logits.device)
            self.eval_topk(torch.arange(len(u_idxs), device=logits.device), idx)
            self._labels_dev = None
        else:
            self.eval_topk(u_idxs, idx)

    def get_results(self, return_raw_results: bool = False):
        keys = list(self._results)
        if keys:                       # one device -> host transfer for all metrics instead of one (and one sync) per metric
            stacked = torch.stack([torch.cat(self._results[k]).float() for k in keys]).cpu().numpy()
            raw = {k: stacked[i] for i, k in enumerate(keys)}
        else:
            raw = {}
        metrics = {k: float(v.mean()) for k, v in raw.items()}
        if getattr(self.config, 'calculate_std', False):
            metrics.update({f'{k}_std': float(v.std()) for k, v in raw.items()})
        if self._topk:
            top = torch.cat(self._topk)
            n_items = self.dataset.n_items_in_split if self.dataset is not None else int(top.max()) + 1
            for k in self._ks:
                if k <= top.shape[1]:
                    # distinct recommended items: mark-and-count over the item range (torch.unique sorts its 2M inputs: ~0.7 ms
                    # per cut-off on c2); empty slots (-1: users with fewer than k scoreable items) are not items
                    ids = top[:, :k].reshape(-1).long()
                    ids = ids[ids >= 0]
                    seen = torch.zeros(max(int(n_items), int(ids.max()) + 1 if ids.numel() else 1), dtype=torch.bool, device=top.device)
                    seen[ids] = True
                    metrics[self._key('coverage', k)] = int(seen.sum()) / n_items
        metrics = {k: metrics[k] for k in sorted(metrics)}
        self._reset()
        return (metrics, raw) if return_raw_results else metrics


def evaluate_recommender_algorithm(alg, eval_loader, evaluator: FullEvaluator, device='cuda', return_raw=False, verbose=False,
                                   scorer: str = 'fp32', user_chunk: Optional[int] = None):
    """eval/eval.py:171-227 (SGD branch :203-222). ``eval_loader`` only has to expose ``dataset`` and ``batch_size``.

    The users are scored in engine-sized chunks, not in the loader's batches: per-user results do not depend on the grouping,
    and the reference's default evaluation batch (256 users) leaves the GPU idle — the fused kernels assign 448 (D <= 128) or 224
    (D = 256) users to a workgroup and every workgroup streams the whole catalogue, so they want >= 57k users per launch (measured on
    c2, 100k users, first kernel: 975 ms with 256-user batches, 37 ms with 8192, 12 ms in one launch); the fp32 path is bounded by the [chunk, items] score
    matrix it materialises. ``user_chunk`` overrides the choice."""
    dataset = eval_loader.dataset
    for attr in ('items_in_split', 'users_in_split', 'exclude_data'):
        if not hasattr(dataset, attr):
            raise ValueError("Dataset underlying loader must be of type 'FullEvaluatorDataset'")
    alg.eval()
    kmax = max(evaluator._ks)
    with torch.no_grad():
        items = torch.as_tensor(np.asarray(dataset.items_in_split)).to(device)
        i_repr = alg.get_item_representations(items)                          # once: [I_s, D] (or a tuple: embeddings, biases, ...)
        plain = torch.is_tensor(i_repr)       # models whose item side is more than one matrix score through their own combine
        i_dev = i_repr.device if plain else i_repr[0].device
        kmax = min(kmax, int(items.shape[0]))
        excl = getattr(dataset, '_excl_dev', None)
        if excl is None or excl[0].device != i_dev:
            excl = _csr_to_device(dataset.exclude_data, i_dev)
            try:
                dataset._excl_dev = excl
            except Exception:
                pass
        users = np.asarray(dataset.users_in_split)
        bs = int(getattr(eval_loader, 'batch_size', 256) or 256)
        if scorer not in ('fp32', 'fp16_fused'):
            raise ValueError(f'unknown scorer {scorer!r}')
        if scorer == 'fp16_fused' and (not plain or kmax > 32 or i_repr.shape[1] not in (64, 128, 256)):
            # the fused kernel keeps at most 32 candidates per user on chip and is built for D in {64, 128, 256}: larger
            # cut-offs (the reference's default evaluator asks for top-100) take the exact fp32 GEMM + radix-select path
            import logging
            logging.info(f'fp16_fused scorer: k={kmax}, item representation {"tuple" if not plain else tuple(i_repr.shape)} outside '
                         f'the fused kernel, using the fp32 path')
            scorer = 'fp32'
        i16 = ops.cast_f16(i_repr) if scorer == 'fp16_fused' else None
        if user_chunk is not None:
            bs = int(user_chunk)
        elif scorer == 'fp16_fused':
            bs = max(bs, 262144)                                    # one launch for up to 256k users (fp16 rows: 64 MB at D = 128)
        else:
            bs = max(bs, min(16384, max(1, (1 << 31) // max(int(items.shape[0]), 1))))      # <= 8 GiB of fp32 scores per chunk
        for s in range(0, len(users), bs):
            u_idxs = torch.from_numpy(users[s:s + bs].astype(np.int64)).to(device)
            u_repr = alg.get_user_representations(u_idxs)
            if scorer == 'fp16_fused' and torch.is_tensor(u_repr):
                _, idx = ops.score_topk_f16(ops.cast_f16(u_repr), i16, kmax, u_idxs, excl[0], excl[1])
            else:
                out = alg.combine_user_item_representations(u_repr, i_repr)
                ops.mask_scores_(out, u_idxs, excl[0], excl[1])
                _, idx = ops.topk_rows(out, kmax)
            evaluator.eval_topk(u_idxs, idx)
        if hasattr(alg, 'check_index_errors'):
            alg.check_index_errors()
    return evaluator.get_results(return_raw_results=return_raw)
