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

import logging
import re
from collections import defaultdict
from typing import Optional, Sequence

import numpy as np
import scipy.sparse as sp
import torch

from . import ops, parallel

SUPPORTED_METRICS = ('ndcg', 'precision', 'recall', 'f_score', 'hitrate', 'coverage')


RANK_METRIC_MAX_KS = 8            # cut-offs per sbr_rank_metrics launch (METRIC_MAX_KS in csrc/topk.hip)


class _Cfg:
    """The fields of the reference's EvalConfig (data/config_classes.py:183-198) with its defaults."""

    def __init__(self, top_k=(1, 3, 5, 10, 20, 50, 100), metrics=SUPPORTED_METRICS, calculate_std=True,
                 calculate_group_metrics=False, user_group_features=None):
        self.top_k, self.metrics, self.calculate_std = list(top_k), list(metrics), calculate_std
        self.calculate_group_metrics, self.user_group_features = calculate_group_metrics, user_group_features


def natural_key(s: str):
    """Sort key of ``natsorted`` (eval/eval.py:160; natsort's default: unsigned integers inside a string compare as numbers,
    so 'ndcg@3' < 'ndcg@10' < 'ndcg@10_std'). natsort itself is not a dependency of this package."""
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', s)]


def _cfg_get(cfg, name, default=None):
    return cfg.get(name, default) if isinstance(cfg, dict) else getattr(cfg, name, default)


def _is_categorical(feature) -> bool:
    t = getattr(getattr(feature, 'feature_definition', None), 'type', None)
    return str(getattr(t, 'value', t)).lower() == 'categorical'


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
        self._user_features = self._determine_user_features()
        self._group_maps = {}            # feature name -> (int32 id -> category map on the device, labels)
        self._warned = False
        self._reset()

    def _determine_user_features(self):
        """eval/eval.py:74-92: the categorical user features for which group-wise metrics are calculated (None: no groups)."""
        if not _cfg_get(self.config, 'calculate_group_metrics', False):
            return None
        ds = self.dataset
        feats = getattr(ds, 'user_features', None)
        if ds is None or feats is None:
            raise ValueError('calculate_group_metrics needs the dataset (its categorical user features define the groups)')
        wanted = _cfg_get(self.config, 'user_group_features')
        if wanted is not None:
            names = getattr(ds, 'user_feature_names', None)
            for feature_name in wanted:
                if feature_name not in (names if names is not None else feats):
                    raise ValueError(f'Dataset does not contain user feature "{feature_name}". '
                                     f'Check config whether features are loaded or set to "None" to use'
                                     f'all available categorical features.')
                if not _is_categorical(feats[feature_name]):
                    raise ValueError(f'User feature "{feature_name}" is not categorical.')
            return list(wanted)
        defs = getattr(ds, 'user_feature_definitions', None)
        if defs is not None:
            return [d.name for d in defs if str(getattr(d.type, 'value', d.type)).lower() == 'categorical']
        # features the model adds itself (sgd_alg.py:2021-2032) are not feature definitions of the dataset
        return [n for n, f in feats.items() if _is_categorical(f) and n not in ('user_embedding', 'interactions')]

    def _reset(self):
        self._results = defaultdict(list)
        self._topk = []
        self._groups = defaultdict(list)     # feature name -> per batch int32 category of every evaluated user

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
        if len(ks) < len(self._ks) and not self._warned:
            self._warned = True
            logging.warning(f'FullEvaluator: cut-offs {[k for k in self._ks if k > kmax]} exceed the {kmax} ranked items per user '
                            f'(catalogue of the split or the scorer\'s list length); their metrics are not reported')
        indptr, indices = self._labels(topk_idx.device)
        topk_idx, u_long = topk_idx.contiguous(), u_idxs.long().contiguous()
        # [3, n_ks, Bu]; the kernel takes up to RANK_METRIC_MAX_KS cut-offs per launch
        m = torch.cat([ops.rank_metrics(topk_idx, u_long, indptr, indices, ks[c:c + RANK_METRIC_MAX_KS])
                       for c in range(0, len(ks), RANK_METRIC_MAX_KS)], dim=1) if ks else None
        for name in (self._user_features or ()):
            gmap = self._group_map(name, topk_idx.device)[0]
            # ids beyond the map (only possible when the dataset does not state n_users): category -1 -> KeyError in get_results
            self._groups[name].append(torch.where(u_long < gmap.numel(), gmap[u_long.clamp(max=max(gmap.numel() - 1, 0))],
                                                  torch.full_like(u_long, -1, dtype=torch.int32)) if gmap.numel() else
                                      torch.full_like(u_long, -1, dtype=torch.int32))
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

    def _group_map(self, name, device):
        """Resident ``user id -> category`` map of one categorical user feature (replaces the per-batch host lookup
        ``user_feature[u_idxs]`` + ``get_labels`` of eval/eval.py:110-113) and the group labels (lower-cased strings)."""
        got = self._group_maps.get(name)
        if got is None or got[0].device != torch.device(device):
            f = self.dataset.user_features[name]
            vals = np.asarray(f.values).astype(np.int64).reshape(-1)
            ids = np.asarray(getattr(f, '_indices', np.arange(len(vals)))).astype(np.int64)
            # one entry per user id the evaluation can name (ids without a value of this feature stay -1 and raise KeyError in
            # get_results, as eval/eval.py:110-113 does on the host) — never a device gather past the end of the map
            n_ids = max(int(ids.max()) + 1 if len(ids) else 0, int(getattr(self.dataset, 'n_users', 0) or 0))
            cat = np.full(n_ids, -1, dtype=np.int32)
            cat[ids] = vals
            uniq = getattr(f, 'unique_values', None)
            n_cat = int(vals.max()) + 1 if len(vals) else 0
            labels = [uniq[c] if uniq is not None and c < len(uniq) else c for c in range(n_cat)]
            labels = [lbl.lower() if isinstance(lbl, str) else lbl for lbl in labels]
            got = self._group_maps[name] = (torch.from_numpy(cat).to(device), labels)
        return got

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
        # group-wise metrics (eval/eval.py:106-119): the same per-user values, restricted to the users of each category of
        # each group feature. Key layout '{evaluator/}{feature}_{label}/{metric}@{k}' as rmet.calculate_for_feature
        # flattens it — rmet is absent offline, so the key layout is parity-unpinned (the values are not: a group's array is
        # the overall array under a mask).
        for name in (self._user_features or ()):
            if not self._groups[name] or not raw:
                continue
            cats = torch.cat(self._groups[name]).cpu().numpy()
            labels = self._group_maps[name][1]
            prefix = f'{self.name}/' if self.name else ''
            for c in np.unique(cats):
                if c < 0:
                    raise KeyError(f'user without a value of group feature "{name}"')
                sel = cats == c
                for k in keys:
                    raw[f'{prefix}{name}_{labels[c]}/{k[len(prefix):]}'] = raw[k][sel]
        metrics = {k: float(v.mean()) for k, v in raw.items()}
        if _cfg_get(self.config, 'calculate_std', False):
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
        metrics = {k: metrics[k] for k in sorted(metrics, key=natural_key)}      # natsorted (eval/eval.py:160)
        self._reset()
        return (metrics, raw) if return_raw_results else metrics


def evaluate_recommender_algorithm(alg, eval_loader, evaluator: FullEvaluator, device='cuda', return_raw=False, verbose=False,
                                   scorer: str = 'fp32', user_chunk: Optional[int] = None, shard_items: bool = False):
    """eval/eval.py:171-227 (SGD branch :203-222). ``eval_loader`` only has to expose ``dataset`` and ``batch_size``.

    The users are scored in engine-sized chunks, not in the loader's batches: per-user results do not depend on the grouping,
    and the reference's default evaluation batch (256 users) leaves the GPU idle — the fused kernel assigns 448 users to a workgroup and every workgroup streams the whole catalogue, so they want >= 57k users per launch (measured on
    c2, 100k users, first kernel: 975 ms with 256-user batches, 37 ms with 8192, 12 ms in one launch); the fp32 path is bounded by the [chunk, items] score
    matrix it materialises. ``user_chunk`` overrides the choice.

    **Item-sharded on request** (``shard_items=True`` under an initialised process group; new: SURVEY.md 8(e), BASELINE configs[4] —
    the reference has no multi-GPU path; ``Trainer.val`` opts in, a rank-0-only evaluation simply does not): EVERY rank of the group
    must make the same call on the same split (the routine runs collectives; the ranks first check that they agree on the split's item
    and user lists and raise otherwise instead of hanging or merging lists of different users). Rank r computes the representations of items [lo_r, hi_r) of the split only and scores every user chunk against them
    (fused kernel with ``item_offset``, or fp32 GEMM + shard-aware mask + top-k); the per-shard ``[Bu, k]`` (score, item position)
    lists are all-gathered and merged exactly (``parallel.all_gather_topk`` -> ``sbr_merge_topk``: score desc, index asc), and every
    rank feeds the merged lists to its evaluator — all ranks return the same metrics as a one-rank evaluation. Models whose item side
    is more than one matrix (biases) score unsharded on every rank."""
    dataset = eval_loader.dataset
    for attr in ('items_in_split', 'users_in_split', 'exclude_data'):
        if not hasattr(dataset, attr):
            raise ValueError("Dataset underlying loader must be of type 'FullEvaluatorDataset'")
    alg.eval()
    kmax = max(evaluator._ks)
    with torch.no_grad():
        items = torch.as_tensor(np.asarray(dataset.items_in_split)).to(device)
        n_split = int(items.shape[0])
        world = parallel.world_size() if (shard_items and parallel.is_distributed()) else 1
        lo, hi = 0, n_split
        sharded = False
        users = np.asarray(dataset.users_in_split)
        if world > 1:
            # all ranks must be evaluating the same split: (items, users, sum of user ids, sum of item ids) agree or nobody shards
            fp = torch.tensor([n_split, len(users), int(users.astype(np.int64).sum()), int(items.sum())], device=device, dtype=torch.int64)
            fp_lo, fp_hi = fp.clone(), fp.clone()
            torch.distributed.all_reduce(fp_lo, op=torch.distributed.ReduceOp.MIN)
            torch.distributed.all_reduce(fp_hi, op=torch.distributed.ReduceOp.MAX)
            if not bool((fp_lo == fp_hi).all()):
                raise ValueError(f'evaluate_recommender_algorithm(shard_items=True): the ranks evaluate different splits '
                                 f'(items / users / id sums: this rank {fp.tolist()}, min {fp_lo.tolist()}, max {fp_hi.tolist()})')
        if world > 1 and (world - 1) * (-(-n_split // world)) < n_split:      # every rank gets a non-empty shard
            rank = torch.distributed.get_rank()
            lo, hi = parallel.item_shard(n_split, rank, world)
            probe = alg.get_item_representations(items[lo:hi])                # this rank's item shard only
            sharded = torch.is_tensor(probe)
            i_repr = probe if sharded else alg.get_item_representations(items)
            if not sharded:
                lo, hi = 0, n_split
        else:
            i_repr = alg.get_item_representations(items)                      # once: [I_s, D] (or a tuple: embeddings, biases, ...)
        plain = torch.is_tensor(i_repr)       # models whose item side is more than one matrix score through their own combine
        i_dev = i_repr.device if plain else i_repr[0].device
        kmax = min(kmax, n_split)
        excl = getattr(dataset, '_excl_dev', None)
        if excl is None or excl[0].device != i_dev:
            excl = _csr_to_device(dataset.exclude_data, i_dev)
            try:
                dataset._excl_dev = excl
            except Exception:
                pass
        bs = int(getattr(eval_loader, 'batch_size', 256) or 256)
        if scorer not in ('fp32', 'fp16_fused'):
            raise ValueError(f'unknown scorer {scorer!r}')
        if scorer == 'fp16_fused' and (not plain or kmax > 32 or i_repr.shape[1] not in (64, 128, 256)):
            # the fused kernel keeps at most 32 candidates per user on chip and is built for D in {64, 128, 256}: larger
            # cut-offs (the reference's default evaluator asks for top-100) take the exact fp32 GEMM + radix-select path
            logging.info(f'fp16_fused scorer: k={kmax}, item representation {"tuple" if not plain else tuple(i_repr.shape)} outside '
                         f'the fused kernel, using the fp32 path')
            scorer = 'fp32'
        i16 = ops.cast_f16(i_repr) if scorer == 'fp16_fused' else None
        if user_chunk is not None:
            bs = int(user_chunk)
        elif scorer == 'fp16_fused':
            bs = max(bs, 262144)                                    # one launch for up to 256k users (fp16 rows: 64 MB at D = 128)
        else:
            bs = max(bs, min(16384, max(1, (1 << 31) // max(hi - lo, 1))))      # <= 8 GiB of fp32 scores per chunk
        for s in range(0, len(users), bs):
            u_idxs = torch.from_numpy(users[s:s + bs].astype(np.int64)).to(device)
            u_repr = alg.get_user_representations(u_idxs)
            if scorer == 'fp16_fused' and torch.is_tensor(u_repr):
                # the split's exclusion mask in the scorer's layout, built once per (user chunk, item shard, D) and kept with the
                # split like the resident CSR it is made from
                cache = getattr(dataset, '_scorer_excl', None)
                if cache is None or cache.get('csr') is not excl:
                    cache = {'csr': excl}
                    try:
                        dataset._scorer_excl = cache
                    except Exception:
                        pass
                # (the key carries a fingerprint of the chunk's user ids: a split whose users_in_split changed between two evaluations
                # must not meet the mask of the old users)
                chunk = users[s:s + bs].astype(np.int64)
                holder = cache.setdefault((s, int(chunk.size), int(chunk[0]), int(chunk[-1]), int(chunk.sum()), lo, hi, int(i16.shape[1])),
                                          ops.ScorerExclusions())
                val, idx = ops.score_topk_f16(ops.cast_f16(u_repr), i16, kmax, u_idxs, excl[0], excl[1], item_offset=lo, exclusions=holder)
            else:
                out = alg.combine_user_item_representations(u_repr, i_repr)
                ops.mask_scores_(out, u_idxs, excl[0], excl[1], item_offset=lo if sharded else None)
                kl = min(kmax, hi - lo)
                val, idx = ops.topk_rows(out, kl)
                if sharded:
                    idx = idx + lo                                  # positions in items_in_split
                    if kl < kmax:                                   # a shard shorter than the list: empty slots behind its items
                        pad = (idx.shape[0], kmax - kl)
                        val = torch.cat([val, torch.full(pad, -float('inf'), device=val.device)], 1)
                        idx = torch.cat([idx, torch.full(pad, -1, device=idx.device, dtype=idx.dtype)], 1)
            if sharded:
                val, idx = parallel.all_gather_topk(val, idx, kmax)
            evaluator.eval_topk(u_idxs, idx)
        if hasattr(alg, 'check_index_errors'):
            alg.check_index_errors()
    return evaluator.get_results(return_raw_results=return_raw)
