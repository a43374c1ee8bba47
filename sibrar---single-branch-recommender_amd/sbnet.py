"""SingleBranchNet / SGDBaseline behind the reference's plugin surface, executed by the HIP engine.

Mirrors (same class names, constructor arguments, method names, error behaviour and state_dict keys):
  algorithms/base_classes.py:87-170   SGDBasedRecommenderAlgorithm
  algorithms/sgd_alg.py:88-123        SGDBaseline
  algorithms/sgd_alg.py:1279-1396     FeatureEmbedding
  algorithms/sgd_alg.py:1764-2006     SingleBranchNetEntity
  algorithms/sgd_alg.py:2009-2144     SingleBranchNet
  train/utils.py:5-13                 general_weight_init

What is different by design (MI355X-first):
  * feature matrices live in HBM (features.DeviceTable); nothing is fetched from the host per batch
    (reference: Feature.__getitem__ numpy round trip per modality per batch, data/Feature.py:159-162);
  * the per-modality Python loop of _get_modality_embeddings becomes one fused front end: slots are counting-sorted by
    modality on the host (the modality draw is host-side anyway), row ids are resolved by one kernel, every modality's
    projector GEMM gathers its rows and scatters its results straight into the [R, C] matrix;
  * the CSR 'interactions' modality is never densified (csr_project kernels); its projector weight keeps the state_dict
    shape [C, n_cols] but is stored column-major so that each nnz reads one contiguous row;
  * modality sampling is the vectorised bit-exact replica of row_wise_sample (sampling.py);
  * the modality ORDER is deterministic (config order) instead of ``list(set(...))`` (hash-seed dependent in the reference).
"""
from __future__ import annotations

import logging
import os
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn
from torch.autograd import Function

from . import ops
from ._lib import call, ptr, stream
from .config import (EmbeddingRegularizationType, FeatureModuleConfig, SingleBranchNetConfig,
                     SingleBranchNetEntityConfig, coerce_net_config, coerce_side_config)
from .features import DeviceTable, HostFeature
from .polylinear import PolyLinear, batch_norm_act, dropout
from .sampling import sample_modalities


def general_weight_init(m):
    """train/utils.py:5-13 — kaiming-uniform(relu) Linear weights, zero biases, N(0, 0.1/dim) embeddings. (The exact type
    check leaves nn.EmbeddingBag at torch's default N(0, 1) with a zero padding row, as in the reference.)"""
    if type(m) == nn.Linear:
        if m.weight.requires_grad:
            torch.nn.init.kaiming_uniform_(m.weight, nonlinearity='relu')
            if hasattr(m, 'bias') and m.bias is not None and m.bias.requires_grad:
                torch.nn.init.constant_(m.bias, 0)
    elif type(m) == nn.Embedding:
        if m.weight.requires_grad:
            torch.nn.init.normal_(m.weight, std=.1 / m.weight.shape[-1])


class SGDBasedRecommenderAlgorithm(nn.Module):
    """algorithms/base_classes.py:87-170."""

    def __init__(self):
        super().__init__()
        self.name = self.__class__

    def forward(self, u_idxs: torch.Tensor, i_idxs: torch.Tensor) -> torch.Tensor:
        u_repr = self.get_user_representations(u_idxs)
        i_repr = self.get_item_representations(i_idxs)
        return self.combine_user_item_representations(u_repr, i_repr)

    def get_user_representations(self, u_idxs):
        raise NotImplementedError

    def get_item_representations(self, i_idxs):
        raise NotImplementedError

    def combine_user_item_representations(self, u_repr, i_repr):
        raise NotImplementedError

    def get_and_reset_other_loss(self) -> Dict:
        return {'reg_loss': torch.zeros(1)}

    def check_index_errors(self):
        """Raises KeyError when any lookup since the last check met a user / item id that has no row in one of its features
        (the reference raises it inside Feature.__getitem__, data/Feature.py:146; here the kernels flag it, substitute row 0
        and keep going, and the host asks at its next synchronisation point: end of an epoch, end of an evaluation)."""
        bad = [name for name, buf in self.named_buffers() if name.endswith('_idx_err') and int(buf.item()) != 0]
        for name, buf in self.named_buffers():
            if name.endswith('_idx_err'):
                buf.zero_()
        if bad:
            raise KeyError('index without feature row in: ' + ', '.join(n.rsplit('.', 1)[0] or '<model>' for n in bad))

    @torch.no_grad()
    def predict(self, u_idxs: torch.Tensor, i_idxs: torch.Tensor) -> torch.Tensor:
        self.eval()
        return self(u_idxs, i_idxs)

    def save_model_to_path(self, path: str):
        path = os.path.join(path, 'model.pth')
        torch.save({k: v.contiguous() for k, v in self.state_dict().items()}, path)
        print('Model Saved')

    def load_model_from_path(self, path: str):
        path = os.path.join(path, 'model.pth')
        self.load_state_dict(torch.load(path, weights_only=True))
        print('Model Loaded')

    @property
    def device(self):
        return next(iter(self.parameters())).device


# ------------------------------------------------------------------------------------------------------------------
# fused modality front end
# ------------------------------------------------------------------------------------------------------------------
class _FrontPlan:
    """Which (feature module, row list, slot list) pairs make up one [R, C] embedding matrix."""
    __slots__ = ('entries', 'rows', 'slots', 'R', 'C')

    def __init__(self, entries, rows, slots, R, C):
        self.entries, self.rows, self.slots, self.R, self.C = entries, rows, slots, R, C


def resolve_rows(idx_flat: torch.Tensor, k: int, slots: torch.Tensor, seg_offsets: List[int], tables: List[DeviceTable],
                 err: Optional[torch.Tensor] = None):
    """One launch: slot -> entity id -> feature-table row (or category id) for all modalities of a plan. ``err`` (device
    int32[1], sticky) is set when an id has no row in a feature's split — the kernel then uses row 0, the host raises KeyError
    (the reference's Feature.__getitem__ dict lookup, data/Feature.py:146) at its next ``check_index_errors``."""
    import ctypes
    n = int(seg_offsets[-1])
    rows = torch.empty(n, device=idx_flat.device, dtype=torch.int32)
    if err is None:
        err = torch.zeros(1, device=idx_flat.device, dtype=torch.int32)
    offs = (ctypes.c_int * len(seg_offsets))(*seg_offsets)
    maps = (ctypes.c_void_p * len(tables))(*[ptr(t.rowmap) for t in tables])
    lens = (ctypes.c_int * len(tables))(*[int(t.rowmap.numel()) if t.rowmap is not None else int(t.n_rows) for t in tables])
    call('sbr_resolve_rows', ptr(idx_flat), k, ptr(slots), n, len(tables), ctypes.cast(offs, ctypes.c_void_p),
         ctypes.cast(maps, ctypes.c_void_p), ctypes.cast(lens, ctypes.c_void_p), ptr(rows), ptr(err), stream())
    return rows, err


class FrontEndFn(Function):
    """All modality front ends of one entity (sgd_alg.py:1934-1978 + FeatureEmbedding.forward :1373-1389) as one autograd
    node. params = concatenation of ``fe.front_params()`` over plan.entries."""

    @staticmethod
    def forward(ctx, plan: _FrontPlan, *params):
        dev = plan.rows.device
        out = torch.empty(plan.R, plan.C, device=dev, dtype=torch.float32)
        hidden_all = []
        pi = 0
        for fe, off, n in plan.entries:
            npar = fe.n_front_params
            p = params[pi:pi + npar]
            pi += npar
            rows = plan.rows[off:off + n]
            slots = plan.slots[off:off + n] if plan.slots is not None else None
            hidden_all.append(fe.front_forward(p, rows, n, out, slots))
        ctx.plan = plan
        ctx.n_params = len(params)
        flat_hidden = [h for hs in hidden_all for h in hs]
        ctx.hidden_counts = [len(hs) for hs in hidden_all]
        ctx.save_for_backward(out, *params, *flat_hidden)
        return out

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        saved = ctx.saved_tensors
        out = saved[0]
        params = saved[1:1 + ctx.n_params]
        hiddens = list(saved[1 + ctx.n_params:])
        dout = dout if (dout.dtype == torch.float32 and dout.is_contiguous()) else dout.float().contiguous()
        grads = []
        pi = hi = 0
        for (fe, off, n), hc in zip(plan.entries, ctx.hidden_counts):
            npar = fe.n_front_params
            p = params[pi:pi + npar]
            pi += npar
            hs = hiddens[hi:hi + hc]
            hi += hc
            rows = plan.rows[off:off + n]
            slots = plan.slots[off:off + n] if plan.slots is not None else None
            grads.extend(fe.front_backward(p, hs, rows, n, out, dout, slots))
        return (None, *grads)


class FeatureEmbedding(nn.Module):
    """algorithms/sgd_alg.py:1279-1396: per-feature front end. dense / CSR features -> PolyLinear projector (activation on
    the output too), categorical -> nn.Embedding, tag -> nn.EmbeddingBag(mean, padding)."""

    # CSR / tag modality, weight gradient: gather form (sbr_csr_project_bwd_gather) or scatter form, by a cost model per column of the
    # gradient — scatter: one float atomic per (slot, entry) at ~3.3 ps; gather: every entry of the matrix read once (~0.6 ps), one
    # atomic per slot, the per-entity workspace zeroed (~0.7 ps per entity) and two more launches. Onion18: the CSR projector takes
    # the gather form from ~2,500 slots (batch 4096: 30,805 slots, 1.25 -> 0.21 ms; batch 256: 1,877 slots, scatter form 76 us
    # against 93), the tag bag from ~9,000. CSR_GATHER_FORCE (tests): True / False overrides the model.
    CSR_GATHER_FORCE = None
    CSR_GATHER_MAX_WS = 1 << 28

    def _gather_pays(self, n: int, nnz: int, n_entities: int, n_cols: int, C: int) -> bool:
        if self.CSR_GATHER_FORCE is not None:
            return bool(self.CSR_GATHER_FORCE)
        adds = n * (nnz / max(n_entities, 1))                 # (slot, entry) pairs of the step
        # ... and the adds that land on ONE gradient row serialise (~30 ns each, whatever C: the columns of a row go in parallel):
        # ML-1M's 18 genre tags take 5,000 adds per row at batch 4096 — 151 us in scatter form
        scatter = max(3.3 * adds, 30e3 * adds / max(n_cols, 1) / C)
        gather = 0.6 * nnz + 3.3 * n + 0.7 * n_entities + 12e6 / C
        return scatter > gather


    def __init__(self, feature, embedding_dim: int = None, pre_embedding_layers: List[int] = None,
                 post_embedding_layers: List[int] = None, activation_fn='relu'):
        super().__init__()
        self._table = DeviceTable(feature)
        self.register_buffer('_idx_err', torch.zeros(1, dtype=torch.int32), persistent=False)   # sticky missing-id flag
        self.kind = self._table.kind
        name = getattr(feature.feature_definition, 'name', '?')
        self._embedding_dim = embedding_dim
        self._activation_fn = activation_fn
        self.output_dim = None

        if embedding_dim is None and self.kind in ('categorical', 'tag'):
            raise ValueError(f'For {self.kind} feature "{name}", the size of its embeddings have to be specified with '
                             f'"embedding_dim"')
        if pre_embedding_layers and self.kind in ('categorical', 'tag'):
            raise ValueError(f'For {self.kind} feature "{name}", using pre-embedding layers would not make any sense '
                             f'(as the input are simple indices).')

        self.pre_embedding_layers = None
        self.embedding_layer = None
        if self.kind in ('dense', 'csr'):
            layer_config = [self._table.dim] + list(pre_embedding_layers or [])
            if embedding_dim is not None:
                layer_config.append(embedding_dim)
            self.output_dim = layer_config[-1]
            if len(layer_config) > 1:
                self.pre_embedding_layers = PolyLinear(layer_config, activation_fn=activation_fn, output_fn=activation_fn)
            else:
                raise NotImplementedError('raw (un-projected) vector features are not part of the SingleBranchNet path')
        elif self.kind == 'categorical':
            self.output_dim = embedding_dim
            self.embedding_layer = nn.Embedding(self._table.n_categories, embedding_dim)
        else:
            self.output_dim = embedding_dim
            self.embedding_layer = nn.EmbeddingBag(self._table.dim + 1, embedding_dim, padding_idx=-1)

        self.post_embedding_layers = None
        if post_embedding_layers:
            cfg = [self.output_dim] + list(post_embedding_layers)
            self.output_dim = cfg[-1]
            self.post_embedding_layers = PolyLinear(cfg, activation_fn=activation_fn, output_fn=activation_fn)

        self.apply(general_weight_init)

        if self.kind == 'csr':
            # keep the state_dict shape [C, n_cols] but store the matrix column-major: W.t() is a contiguous [n_cols, C]
            lin = self.pre_embedding_layers.layers.linear_0
            w = lin.weight.data
            lin.weight = nn.Parameter(w.t().contiguous().t())
        self._act = ops.act_code(activation_fn)
        self._colsum_ws = {}            # layer -> column-reduction workspace of its folded bias gradient (fused step)
        self._csr_ws = None             # [n_entities, C] per-entity gradient rows of a CSR modality (gather form of its backward)

    # -- front-end protocol used by FrontEndFn ---------------------------------------------------------------------------
    @property
    def front_dim(self) -> int:
        """width of what the front end writes (before post_embedding_layers)"""
        if self.pre_embedding_layers is not None:
            return self.pre_embedding_layers.layer_config[-1]
        return self._embedding_dim

    def front_params(self) -> List[torch.Tensor]:
        if self.pre_embedding_layers is not None:
            ps = []
            for i in range(self.pre_embedding_layers.n_layers):
                lin = getattr(self.pre_embedding_layers.layers, f'linear_{i}')
                ps += [lin.weight, lin.bias]
            return ps
        return [self.embedding_layer.weight]

    @property
    def n_front_params(self) -> int:
        return 2 * self.pre_embedding_layers.n_layers if self.pre_embedding_layers is not None else 1

    @staticmethod
    def _colmajor(w: torch.Tensor) -> torch.Tensor:
        wt = w.t()
        return wt if wt.is_contiguous() else wt.contiguous()

    def front_forward(self, p, rows, n, out, slots):
        """Writes act(...) of this modality's n rows into out[slots]; returns the saved intermediate activations."""
        t = self._table
        st = stream()
        if self.kind == 'categorical':
            call('sbr_gather_rows', ptr(p[0]), p[0].stride(0), ptr(rows), ptr(out), out.stride(0), ptr(slots), n,
                 out.shape[1], st)
            return []
        if self.kind == 'tag':
            call('sbr_bag_mean_fwd', ptr(p[0]), p[0].stride(0), ptr(t.tags), t.T, t.pad, ptr(rows), ptr(out), out.stride(0),
                 ptr(slots), n, out.shape[1], st)
            return []
        L = len(p) // 2
        hidden = []
        h = None
        for l in range(L):
            W, b = p[2 * l], p[2 * l + 1]
            last = l == L - 1
            if last:
                dst, cidx = out, slots
            else:
                dst, cidx = torch.empty(n, W.shape[0], device=out.device, dtype=torch.float32), None
                hidden.append(dst)
            if l == 0 and self.kind == 'csr':
                wt = self._colmajor(W)
                call('sbr_csr_project_fwd', ptr(t.indptr), ptr(t.indices), ptr(t.data), ptr(wt), wt.stride(0), ptr(b), ptr(rows),
                     ptr(dst), dst.stride(0), ptr(cidx), n, W.shape[0], self._act, st)
            elif l == 0:
                ops.linear_nt(t.values, W, b, self._act, a_idx=rows, out=dst, c_idx=cidx, n_rows=n)
            else:
                ops.linear_nt(h, W, b, self._act, out=dst, c_idx=cidx, n_rows=n)
            h = dst
        return hidden

    def front_backward(self, p, hidden, rows, n, out, dout, slots, grad_out=None, pending=None, tn=None):
        """Gradients of this modality's parameters. ``grad_out`` (optional): tensors to write into instead of fresh ones —
        lookup-table gradients are ACCUMULATED into them (they must be zero-initialised), dense ones are overwritten.
        ``pending`` (optional list, needs ``grad_out``): the bias gradients are left pending as column sums folded into the
        activation-derivative kernels — (workspace, bias gradient) pairs are appended and the caller completes them with
        ``ops.colred_finish`` (one launch for all layers of a step instead of two per bias). ``tn`` (optional ``ops.DeferredTN``,
        needs ``grad_out``): the weight-gradient products only write their split-K slabs; the caller sums them with ``tn.finish()``."""
        t = self._table
        st = stream()
        if self.kind == 'categorical':
            dW = torch.zeros_like(p[0]) if grad_out is None else grad_out[0]
            call('sbr_scatter_add_rows', ptr(dout), dout.stride(0), ptr(slots), ptr(rows), ptr(dW), dW.stride(0), n,
                 dout.shape[1], st)
            return [dW]
        if self.kind == 'tag':
            dW = torch.zeros_like(p[0]) if grad_out is None else grad_out[0]
            C = dout.shape[1]
            if (C % 4 == 0 and C <= 1024 and dW.stride(0) % 4 == 0 and dW.data_ptr() % 16 == 0 and t.n_rows * C <= self.CSR_GATHER_MAX_WS
                    and self._gather_pays(n, t.n_entries(), t.n_rows, dW.shape[0], C)):
                # the mean over a tag list is a product with X[entity, tag] = 1 / (tags of the entity): the gather form of the CSR
                # projector's backward pass (per-entity sums, then every tag gathers its entities' rows) serves it as it is
                ti, tj, tv = t.transposed(dW.shape[0])
                ws = self._csr_ws
                if ws is None or ws.shape != (t.n_rows, C) or ws.device != dout.device:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError('the tag gradient workspace must exist before a step is captured (run one plain step first)')
                    ws = self._csr_ws = torch.empty(t.n_rows, C, device=dout.device, dtype=torch.float32)
                call('sbr_csr_project_bwd_gather', ptr(ti), ptr(tj), ptr(tv), ptr(dout), dout.stride(0), ptr(slots), ptr(rows), n, ptr(ws), C,
                     t.n_rows, ptr(dW), dW.stride(0), dW.shape[0], C, st)
                return [dW]
            call('sbr_bag_mean_bwd', ptr(dout), dout.stride(0), ptr(slots), ptr(t.tags), t.T, t.pad, ptr(rows), ptr(dW),
                 dW.stride(0), n, C, st)
            return [dW]
        L = len(p) // 2
        grads = [None] * (2 * L)
        go = grad_out if grad_out is not None else [None] * (2 * L)

        def dz_of(dy, y, l, idx=None, n_rows=None):
            """dy * act'(y) (rows gathered through idx) and the bias gradient of layer l: folded or by its own reduction"""
            C = dy.shape[1]
            if pending is not None and go[2 * l + 1] is not None and ops.colsum_supported(C) and dy.stride(0) % 4 == 0:
                ws = self._colsum_ws.get(l)
                if ws is None:
                    ws = self._colsum_ws[l] = ops.new_colsum_ws(dy.device, C)
                dz_ = ops.act_grad_colsum(dy, y, self._act, ws, idx=idx, n_rows=n_rows)
                pending.append((ws, go[2 * l + 1]))
                grads[2 * l + 1] = go[2 * l + 1]
                return dz_
            dz_ = ops.act_grad(dy, y, self._act, idx=idx, n_rows=n_rows)
            grads[2 * l + 1] = ops.colsum(dz_, out=go[2 * l + 1])
            return dz_

        dz = dz_of(dout, out, L - 1, idx=slots, n_rows=n)                     # [n, C] compact
        for l in range(L - 1, -1, -1):
            W = p[2 * l]
            if l == 0 and self.kind == 'csr':
                # the projector weight is column-major: its gradient is accumulated in the same [n_cols, C] layout
                if go[0] is not None:
                    dWt = go[0].t()
                    assert dWt.is_contiguous()
                else:
                    dWt = torch.zeros(W.shape[1], W.shape[0], device=W.device, dtype=torch.float32)
                C = W.shape[0]
                if (C % 4 == 0 and C <= 1024 and dWt.stride(0) % 4 == 0 and dWt.data_ptr() % 16 == 0 and t.n_rows * C <= self.CSR_GATHER_MAX_WS
                        and self._gather_pays(n, t.n_entries(), t.n_rows, W.shape[1], C)):
                    # many slots: add the slot gradients up per entity first, then every feature column gathers the rows of the
                    # entities that have it (the forward kernel over the transposed matrix; no atomics on the weight gradient)
                    ti, tj, tv = t.transposed()
                    ws = self._csr_ws
                    if ws is None or ws.shape != (t.n_rows, C) or ws.device != dz.device:
                        if torch.cuda.is_current_stream_capturing():
                            raise RuntimeError('the CSR gradient workspace must exist before a step is captured (run one plain step first)')
                        ws = self._csr_ws = torch.empty(t.n_rows, C, device=dz.device, dtype=torch.float32)
                    call('sbr_csr_project_bwd_gather', ptr(ti), ptr(tj), ptr(tv), ptr(dz), dz.stride(0), None, ptr(rows), n, ptr(ws), C, t.n_rows,
                         ptr(dWt), dWt.stride(0), W.shape[1], C, st)
                else:
                    call('sbr_csr_project_bwd', ptr(t.indptr), ptr(t.indices), ptr(t.data), ptr(dz), dz.stride(0), ptr(rows),
                         ptr(dWt), dWt.stride(0), n, C, st)
                grads[0] = dWt.t()
            elif l == 0:
                if tn is not None and go[0] is not None:
                    grads[0] = tn.matmul_tn((id(self), 0), dz, t.values, b_idx=rows, n_rows=n, out=go[0])
                else:
                    grads[0] = ops.matmul_tn(dz, t.values, b_idx=rows, n_rows=n, out=go[0])
            else:
                if tn is not None and go[2 * l] is not None:
                    grads[2 * l] = tn.matmul_tn((id(self), l), dz, hidden[l - 1], n_rows=n, out=go[2 * l])
                else:
                    grads[2 * l] = ops.matmul_tn(dz, hidden[l - 1], n_rows=n, out=go[2 * l])
                dh = ops.matmul_nn(dz, W if W.stride(1) == 1 else W.contiguous())
                dz = dz_of(dh, hidden[l - 1], l - 1)
        return grads

    # -- stand-alone use (plain user / item side of SingleBranchNet) -------------------------------------------------------
    def forward(self, indices: torch.Tensor):
        if not indices.is_cuda:
            raise RuntimeError('FeatureEmbedding (HIP engine) needs CUDA(HIP) index tensors')
        flat = indices.reshape(-1).long().contiguous()
        n = flat.numel()
        slots = torch.arange(n, device=flat.device, dtype=torch.int32)
        rows, err = resolve_rows(flat, 1, slots, [0, n], [self._table], self._idx_err)
        plan = _FrontPlan([(self, 0, n)], rows, None, n, self.front_dim)
        x = FrontEndFn.apply(plan, *self.front_params())
        if self.post_embedding_layers is not None:
            x = self.post_embedding_layers(x)
        if self.kind == 'categorical':
            return x                         # categorical features come back flat (Feature.py:152-155)
        return x.reshape(*indices.shape, x.shape[-1])

    @classmethod
    def build_from_conf(cls, config: FeatureModuleConfig, feature):
        conf = config.to_dict()
        conf.pop('feature_name')
        return cls(feature, **conf)


AGGREGATION_FUNCTIONS = {'mean': 0, 'max': 1}


class SingleBranchNetEntity(nn.Module):
    """algorithms/sgd_alg.py:1764-2006."""

    def __init__(self, entity_name: str, features: Dict[str, object], entity_config: SingleBranchNetEntityConfig,
                 shared_common_dim: int, val_interactions_available: bool = True,
                 train_modality_order: Optional[List[str]] = None, eval_modality_order: Optional[List[str]] = None):
        super().__init__()
        entity_config = coerce_side_config(entity_config)
        self.features = features
        self.entity_name = entity_name
        self.entity_config = entity_config
        self.output_dim = shared_common_dim
        self.val_interactions_available = val_interactions_available
        self.register_buffer('_idx_err', torch.zeros(1, dtype=torch.int32), persistent=False)      # sticky missing-id flag

        if len(entity_config.features) == 0:
            raise ValueError('SingleBranchEntity requires at least one feature.')

        self.train_modalities = self._get_modalities(train=True)
        self.eval_modalities = self._get_modalities(train=False)

        not_available_modalities = self.train_modalities - set(features.keys())
        if len(not_available_modalities) > 0:
            raise ValueError(f'Features for modalities {not_available_modalities} are not available!')
        not_available_definitions = self.train_modalities - set(f.feature_name for f in entity_config.features)
        if len(not_available_definitions) > 0:
            raise ValueError(f'Network definitions for modalities {not_available_definitions} are not available!')

        # deterministic modality order: configuration order (the reference iterates a set)
        cfg_order = [f.feature_name for f in entity_config.features]
        self.train_modality_order = list(train_modality_order) if train_modality_order is not None \
            else [m for m in cfg_order if m in self.train_modalities]
        self.eval_modality_order = list(eval_modality_order) if eval_modality_order is not None \
            else [m for m in cfg_order if m in self.eval_modalities]
        if set(self.train_modality_order) != self.train_modalities or set(self.eval_modality_order) != self.eval_modalities:
            raise ValueError('explicit modality orders must be permutations of the train / eval modality sets')

        self.modality_modules = nn.ModuleDict()
        for f in entity_config.features:
            if f.feature_name not in self.train_modalities:
                continue
            feature_conf = FeatureModuleConfig(feature_name=f.feature_name, embedding_dim=entity_config.common_modality_dim,
                                               pre_embedding_layers=f.feature_hidden_layers,
                                               activation_fn=entity_config.activation_fn)
            self.modality_modules[f.feature_name] = FeatureEmbedding.build_from_conf(feature_conf, features[f.feature_name])

        sb_net_layers = []
        if entity_config.single_branch_input_dropout is not None:
            sb_net_layers.append(nn.Dropout(entity_config.single_branch_input_dropout))
        apply_batch_norm_every = entity_config.apply_batch_norm_every if entity_config.apply_batch_normalization else 0
        sb_net_layers.append(PolyLinear(
            [entity_config.common_modality_dim] + list(entity_config.single_branch_hidden_layers) + [self.output_dim],
            activation_fn=entity_config.activation_fn,
            output_fn=entity_config.activation_fn if entity_config.apply_output_activation else None,
            apply_batch_norm_every=apply_batch_norm_every))
        self._poly_index = len(sb_net_layers) - 1
        self._trailing_bn = entity_config.apply_batch_normalization and entity_config.apply_batch_norm_every == 0
        if self._trailing_bn:
            sb_net_layers.append(nn.BatchNorm1d(self.output_dim))
        self.sb_net = nn.Sequential(*sb_net_layers)

        if entity_config.aggregation_fn not in AGGREGATION_FUNCTIONS:
            raise ValueError(f'Aggregation function "{entity_config.aggregation_fn}" is not supported.')
        self._agg_mode = AGGREGATION_FUNCTIONS[entity_config.aggregation_fn]

        self._reg_type = entity_config.embedding_regularization_type
        if not isinstance(self._reg_type, EmbeddingRegularizationType):
            self._reg_type = EmbeddingRegularizationType(getattr(self._reg_type, 'value', self._reg_type))
        self.regularization_loss = None
        self._rng = np.random.default_rng(entity_config.sampling_seed)
        self.last_modalities = None          # position array of the last forward (for tests / logging)

    # -- modality bookkeeping -------------------------------------------------------------------------------------------
    def _get_modalities(self, train=True):
        available_mods = {f.feature_name for f in self.entity_config.features}
        if train:
            mods = set(self.entity_config.train_modalities or available_mods)
        else:
            train_mods = self._get_modalities(train=True)
            if self.entity_config.eval_modalities is not None:
                for m in self.entity_config.eval_modalities:
                    if m not in train_mods:
                        raise ValueError(f'Cannot use modality "{m}" during evaluation, if it is not used during training.')
            mods = set(self.entity_config.eval_modalities or train_mods)
            if not self.val_interactions_available:
                mods.discard('interactions')
        if len(mods) == 0:
            raise ValueError(f'No single modality is available during {"training" if train else "evaluation"}: '
                             f'There are either no modalities specified or no interactions are available)')
        return mods

    def _sample_modalities(self, shape) -> Tuple[np.ndarray, List[str]]:
        """-> (positions int8 [n_slots, k], the ordered modality list the positions refer to). sgd_alg.py:1904-1932."""
        n_slots = int(np.prod(shape)) if len(shape) else 1
        if self.training:
            order = self.train_modality_order
            pos = sample_modalities(self._rng, order, n_slots, self._reg_type.value, self.entity_config.central_modality)
            return pos, order
        order = self.eval_modality_order
        pos = np.tile(np.arange(len(order), dtype=np.int8), (n_slots, 1))
        return pos, order

    def modality_names(self, pos: np.ndarray, order: List[str]) -> np.ndarray:
        return np.array(order)[pos]

    # -- forward ---------------------------------------------------------------------------------------------------------
    def _front(self, indices: torch.Tensor, pos: np.ndarray, order: List[str]) -> torch.Tensor:
        k = pos.shape[1]
        flat_mods = pos.reshape(-1)
        R = flat_mods.size
        # counting sort of the R slots by modality (stable): slot lists per modality, concatenated
        order_idx = np.argsort(flat_mods, kind='stable').astype(np.int32)
        counts = np.bincount(flat_mods, minlength=len(order))
        dev = indices.device
        from ._lib import to_device
        slots = to_device(torch.from_numpy(order_idx), dev)
        entries, tables, offs = [], [], [0]
        for m, c in enumerate(counts.tolist()):
            if c == 0:
                continue
            fe = self.modality_modules[order[m]]
            entries.append((fe, offs[-1], c))
            tables.append(fe._table)
            offs.append(offs[-1] + c)
        idx_flat = indices.reshape(-1).long().contiguous()
        rows, err = resolve_rows(idx_flat, k, slots, offs, tables, self._idx_err)
        plan = _FrontPlan(entries, rows, slots, R, self.entity_config.common_modality_dim)
        params = [p for fe, _, _ in entries for p in fe.front_params()]
        return FrontEndFn.apply(plan, *params)

    def _embed(self, indices, pos, order):
        x = self._front(indices, pos, order)                       # [R, C]
        if self.entity_config.normalize_single_branch_input:
            x = ops.L2NormalizeFn.apply(x)
        x = dropout(x, self.entity_config.single_branch_input_dropout, self.training)
        x = self.sb_net[self._poly_index](x)
        if self._trailing_bn:
            x = batch_norm_act(self.sb_net[self._poly_index + 1], x, 0, self.training)
        return x.view(-1, pos.shape[1], self.output_dim)             # [S, k, D]

    def forward(self, indices: torch.Tensor, modalities: Optional[np.ndarray] = None):
        """``modalities`` (optional): explicit array of modality NAMES of shape indices.shape + (k,), overriding the draw
        (used by the golden-vector tests that replay the reference's recorded decisions)."""
        if not indices.is_cuda:
            raise RuntimeError('SingleBranchNetEntity (HIP engine) needs CUDA(HIP) index tensors')
        if modalities is not None:
            if tuple(indices.shape) != tuple(modalities.shape[:-1]):
                raise ValueError('Shape of indices and modalities (up to the last dimension) does not match.')
            order = self.train_modality_order if self.training else self.eval_modality_order
            lut = {m: i for i, m in enumerate(order)}
            pos = np.vectorize(lut.__getitem__, otypes=[np.int8])(np.asarray(modalities)).reshape(-1, modalities.shape[-1])
        else:
            pos, order = self._sample_modalities(tuple(indices.shape))
        self.last_modalities = (pos, order)
        e = self._embed(indices, pos, order)
        if self.training:
            self.compute_reg_losses(e, indices.shape)
        k = e.shape[1]
        out = e[:, 0] if k == 1 else ops.AggregateFn.apply(e, self._agg_mode)
        return out.reshape(*indices.shape, self.output_dim)

    def compute_reg_losses(self, e: torch.Tensor, index_shape):
        if self._reg_type == EmbeddingRegularizationType.NoRegularization:
            self.regularization_loss = None
            return
        if e.shape[1] != 2:
            raise SystemError('second last dimension of embeddings should be of size 2')
        # InfoNCE over the last index dimension: [B, N] -> B groups of N; [B] -> one group of B (in-batch)
        N = int(index_shape[-1])
        G = int(np.prod(index_shape[:-1])) if len(index_shape) > 1 else 1
        self.regularization_loss = ops.InfoNCEFn.apply(e, float(self.entity_config.regularization_temperature), True, G, N)

    def _zero_loss(self):
        return torch.tensor([0.], device=next(iter(self.parameters())).device)

    def get_and_reset_other_loss(self) -> Dict:
        loss = self.regularization_loss if self.regularization_loss is not None else self._zero_loss()
        loss = loss * self.entity_config.regularization_weight
        self.regularization_loss = None
        return {'reg_loss': loss}


class SingleBranchNet(SGDBasedRecommenderAlgorithm):
    """algorithms/sgd_alg.py:2009-2144."""

    def __init__(self, config, dataset, modality_orders: Optional[dict] = None):
        super().__init__()
        config = coerce_net_config(config)
        self.config = config
        orders = modality_orders or {}

        user_features = dataset.user_features
        user_features['interactions'] = HostFeature('interactions', 'csr', dataset.user_sampling_matrix_train)
        user_features['user_embedding'] = HostFeature('user_embedding', 'categorical', np.arange(dataset.n_users),
                                                      n_categories=dataset.n_users)
        self.is_user_sb_module = config.is_user_sb_module
        if self.is_user_sb_module:
            self.user_embedding_module = SingleBranchNetEntity(
                'user', user_features, config.user, config.shared_common_dim,
                val_interactions_available=not dataset.is_cold_start_user,
                train_modality_order=orders.get('user_train'), eval_modality_order=orders.get('user_eval'))
        else:
            user_conf = config.user
            if user_conf.embedding_dim == -1:
                user_conf.embedding_dim = config.shared_common_dim
            self.user_embedding_module = FeatureEmbedding.build_from_conf(user_conf, user_features[user_conf.feature_name])

        item_features = dataset.item_features
        item_features['interactions'] = HostFeature('interactions', 'csr', dataset.item_sampling_matrix_train)
        item_features['item_embedding'] = HostFeature('item_embedding', 'categorical', np.arange(dataset.n_items),
                                                      n_categories=dataset.n_items)
        self.is_item_sb_module = config.is_item_sb_module
        if self.is_item_sb_module:
            self.item_embedding_module = SingleBranchNetEntity(
                'item', item_features, config.item, config.shared_common_dim,
                val_interactions_available=not dataset.is_cold_start_item,
                train_modality_order=orders.get('item_train'), eval_modality_order=orders.get('item_eval'))
        else:
            item_conf = config.item
            if item_conf.embedding_dim == -1:
                item_conf.embedding_dim = config.shared_common_dim
            self.item_embedding_module = FeatureEmbedding.build_from_conf(item_conf, item_features[item_conf.feature_name])

        self.name = 'SingleBranchNet'
        logging.info(f'Built {self.name} module')

    def get_user_representations(self, u_idxs: torch.Tensor, modalities=None):
        if self.is_user_sb_module:
            return self.user_embedding_module(u_idxs, modalities)
        return self.user_embedding_module(u_idxs)

    def get_item_representations(self, i_idxs: torch.Tensor, modalities=None):
        if self.is_item_sb_module:
            return self.item_embedding_module(i_idxs, modalities)
        return self.item_embedding_module(i_idxs)

    def combine_user_item_representations(self, u_repr, i_repr):
        if i_repr.ndim == 2:
            return ops.ScoreAllFn.apply(u_repr, i_repr)              # einsum('be,ce->bc')
        return ops.ScoreDotFn.apply(u_repr, i_repr)                  # einsum('be,bce->bc')

    def forward(self, u_idxs, i_idxs, user_modalities=None, item_modalities=None):
        u_repr = self.get_user_representations(u_idxs, user_modalities)
        i_repr = self.get_item_representations(i_idxs, item_modalities)
        return self.combine_user_item_representations(u_repr, i_repr)

    def get_and_reset_other_loss(self) -> Dict:
        losses = {'reg_loss': torch.tensor([0.]).to(self.device)}
        if self.is_user_sb_module:
            r = self.user_embedding_module.get_and_reset_other_loss()
            losses['reg_loss'] = losses['reg_loss'] + r['reg_loss']
            losses.update({f'user_{k}': v for k, v in r.items()})
        if self.is_item_sb_module:
            r = self.item_embedding_module.get_and_reset_other_loss()
            losses['reg_loss'] = losses['reg_loss'] + r['reg_loss']
            losses.update({f'item_{k}': v for k, v in r.items()})
        return losses

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return SingleBranchNet(SingleBranchNetConfig.from_dict(conf) if isinstance(conf, dict) else conf, dataset)


class SGDBaseline(SGDBasedRecommenderAlgorithm):
    """algorithms/sgd_alg.py:88-123 — global + user + item bias."""

    def __init__(self, n_users: int, n_items: int):
        super().__init__()
        self.n_users, self.n_items = n_users, n_items
        self.user_bias = nn.Embedding(n_users, 1)
        self.item_bias = nn.Embedding(n_items, 1)
        self.global_bias = nn.Parameter(torch.zeros(1), requires_grad=True)
        self.apply(general_weight_init)
        self.name = 'SGDBaseline'

    def get_user_representations(self, u_idxs):
        return ops.LookupFn.apply(self.user_bias.weight, u_idxs)

    def get_item_representations(self, i_idxs):
        return ops.LookupFn.apply(self.item_bias.weight, i_idxs).squeeze()

    def combine_user_item_representations(self, u_repr, i_repr):
        # u_repr [B, 1], i_repr [B, N] (training) or [I] (evaluation, eval/eval.py:209-217): gathered values, no further lookup
        N = i_repr.shape[-1]
        B = u_repr.shape[0]
        base = i_repr.contiguous() if i_repr.ndim == 2 else None
        return ops.BiasScoreFn.apply(base, u_repr.reshape(-1), None if i_repr.ndim == 2 else i_repr.contiguous(),
                                     self.global_bias, None, None, B, N)

    def forward(self, u_idxs, i_idxs):
        if not i_idxs.is_cuda:
            raise RuntimeError('SGDBaseline (HIP engine) needs CUDA(HIP) index tensors')
        if i_idxs.ndim == 2:                               # one fused kernel forward, one backward
            B, N = i_idxs.shape
            return ops.BiasScoreFn.apply(None, self.user_bias.weight.view(-1), self.item_bias.weight.view(-1), self.global_bias,
                                         u_idxs, i_idxs, B, N)
        return super().forward(u_idxs, i_idxs)

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return SGDBaseline(dataset.n_users, dataset.n_items)


class SGDMatrixFactorization(SGDBasedRecommenderAlgorithm):
    """algorithms/sgd_alg.py:126-200 — matrix factorisation trained by gradient descent (a sibling model that
    runs on the hot path's kernels): embedding lookups (``sbr_gather_rows`` / dense ``sbr_scatter_add_rows``), the per-slot dot
    product of the SingleBranchNet scorer (``sbr_score_dot_*``) or the all-pairs MFMA GEMM in evaluation, bias terms
    (``sbr_bias_score_*``). state_dict keys as in the reference: user_embeddings.weight, item_embeddings.weight,
    [user_bias.weight], [item_bias.weight], [global_bias]."""

    def __init__(self, n_users: int, n_items: int, embedding_dim: int = 100, use_user_bias: bool = False,
                 use_item_bias: bool = False, use_global_bias: bool = False):
        super().__init__()
        self.n_users, self.n_items, self.embedding_dim = n_users, n_items, embedding_dim
        self.use_user_bias, self.use_item_bias, self.use_global_bias = use_user_bias, use_item_bias, use_global_bias
        self.user_embeddings = nn.Embedding(n_users, embedding_dim)
        self.item_embeddings = nn.Embedding(n_items, embedding_dim)
        if use_user_bias:
            self.user_bias = nn.Embedding(n_users, 1)
        if use_item_bias:
            self.item_bias = nn.Embedding(n_items, 1)
        self.apply(general_weight_init)
        if use_global_bias:
            self.global_bias = nn.Parameter(torch.zeros(1), requires_grad=True)
        self.name = 'SGDMatrixFactorization'

    def get_user_representations(self, u_idxs):
        emb = ops.LookupFn.apply(self.user_embeddings.weight, u_idxs)
        if self.use_user_bias:
            return emb, ops.LookupFn.apply(self.user_bias.weight, u_idxs)
        return emb

    def get_item_representations(self, i_idxs):
        emb = ops.LookupFn.apply(self.item_embeddings.weight, i_idxs)
        if self.use_item_bias:
            return emb, ops.LookupFn.apply(self.item_bias.weight, i_idxs).squeeze()
        return emb

    def _check_user_bias(self):
        if self.use_user_bias:
            # the reference adds u_bias[:, None] ([B, 1, 1]) in place to the [B, N] scores (sgd_alg.py:190): a RuntimeError
            raise RuntimeError("output with shape [B, N] doesn't match the broadcast shape [B, B, N] "
                               '(use_user_bias: the reference raises here, algorithms/sgd_alg.py:190)')

    def combine_user_item_representations(self, u_repr, i_repr):
        u_embed = u_repr[0] if isinstance(u_repr, tuple) else u_repr
        i_embed, i_bias = i_repr if isinstance(i_repr, tuple) else (i_repr, None)
        if i_embed.ndim == 3 and i_embed.shape[0] == 1:    # [1, I, D]: broadcast over users like the reference's product
            i_embed = i_embed[0]
            i_bias = i_bias.reshape(-1) if i_bias is not None else None
        out = (ops.ScoreDotFn if i_embed.ndim == 3 else ops.ScoreAllFn).apply(u_embed, i_embed)
        self._check_user_bias()
        if i_bias is None and not self.use_global_bias:
            return out
        B, N = out.shape
        gb = self.global_bias if self.use_global_bias else None
        if i_bias is None or i_bias.ndim == 2:             # per-slot biases are already [B, N]: part of the base
            base = out if i_bias is None else out + i_bias
            return ops.BiasScoreFn.apply(base, None, None, gb, None, None, B, N)
        return ops.BiasScoreFn.apply(out, None, i_bias.contiguous(), gb, None, None, B, N)

    def forward(self, u_idxs, i_idxs):
        if not i_idxs.is_cuda:
            raise RuntimeError('SGDMatrixFactorization (HIP engine) needs CUDA(HIP) index tensors')
        if i_idxs.ndim != 2:
            return super().forward(u_idxs, i_idxs)
        # training path: lookups, per-slot dot, bias terms — every step a HIP kernel with its hand-written backward
        u_embed = ops.LookupFn.apply(self.user_embeddings.weight, u_idxs)
        i_embed = ops.LookupFn.apply(self.item_embeddings.weight, i_idxs)
        out = ops.ScoreDotFn.apply(u_embed, i_embed)
        self._check_user_bias()
        if not (self.use_item_bias or self.use_global_bias):
            return out
        B, N = i_idxs.shape
        return ops.BiasScoreFn.apply(out, None, self.item_bias.weight.view(-1) if self.use_item_bias else None,
                                     self.global_bias if self.use_global_bias else None, None, i_idxs, B, N)

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return SGDMatrixFactorization(dataset.n_users, dataset.n_items, conf['embedding_dim'], conf['use_user_bias'],
                                      conf['use_item_bias'], conf['use_global_bias'])


class _FeatureMatrixFactorization(SGDMatrixFactorization):
    """Shared body of ItemFeatureMatrixFactorization / UserFeatureMatrixFactorization (algorithms/sgd_alg.py:1399-1614): matrix
    factorisation whose item (user) embeddings are pulled towards a projection of a content feature by an InfoNCE term. Reuses
    the path's kernels: FeatureEmbedding front end (gathered-row MFMA GEMM / lookups), InfoNCE forward + gradient, the
    modality-mean kernel for ``aggregate_for_rec``, the MF scorer. As in the reference, ``get_and_reset_other_loss`` returns the
    contrastive loss UNSCALED (``lambda_content`` is stored but never applied, sgd_alg.py:1491-1497)."""

    _side = None      # 'item' | 'user'

    def __init__(self, dataset, feature_name: str, aggregate_for_rec: bool = False, lambda_content: float = 0.0001,
                 temperature: float = 0.1, embedding_loss_aggregator: str = 'mean', intermediate_layers=None,
                 embedding_dim: int = 100, use_user_bias: bool = False, use_item_bias: bool = False,
                 use_global_bias: bool = False):
        super().__init__(dataset.n_users, dataset.n_items, embedding_dim, use_user_bias, use_item_bias, use_global_bias)
        from .losses import InfoNCE
        self.feature_name = feature_name
        self.aggregate_for_rec = aggregate_for_rec
        self.lambda_content = lambda_content
        features = dataset.item_features if self._side == 'item' else dataset.user_features
        self.embedding_net = FeatureEmbedding(feature=features[feature_name], pre_embedding_layers=intermediate_layers,
                                              embedding_dim=embedding_dim)
        self.emb_loss_fn = InfoNCE(temperature, embedding_loss_aggregator)
        self.emb_loss = 0.

    @staticmethod
    def _mean2(a, b):
        """torch.stack([a, b]).mean(0) through the modality-mean kernel."""
        D = a.shape[-1]
        e = torch.stack([a.reshape(-1, D), b.reshape(-1, D)], dim=1)
        return ops.AggregateFn.apply(e, 0).view(a.shape)

    def compute_reg_losses(self, profile_embs, content_embs):
        self.emb_loss = self.emb_loss_fn(profile_embs, content_embs)

    def get_and_reset_other_loss(self) -> Dict:
        emb_loss = self.emb_loss
        self.emb_loss = 0
        return {'reg_loss': emb_loss}


class ItemFeatureMatrixFactorization(_FeatureMatrixFactorization):
    """algorithms/sgd_alg.py:1399-1505."""
    _side = 'item'

    def forward(self, u_idxs, i_idxs):
        u_repr = self.get_user_representations(u_idxs)
        i_repr = self.get_item_representations(i_idxs)
        dots = self.combine_user_item_representations(u_repr, i_repr)
        self.compute_reg_losses(i_repr[0], i_repr[1])
        return dots

    def get_item_representations(self, i_idxs):
        profile = ops.LookupFn.apply(self.item_embeddings.weight, i_idxs)
        content = self.embedding_net(i_idxs)
        if self.use_item_bias:
            return profile, content, ops.LookupFn.apply(self.item_bias.weight, i_idxs).squeeze()
        return profile, content

    def combine_user_item_representations(self, u_repr, i_repr):
        i_embed = self._mean2(i_repr[0], i_repr[1]) if self.aggregate_for_rec else i_repr[0]
        if self.use_item_bias:
            return super().combine_user_item_representations(u_repr, (i_embed, i_repr[-1]))
        return super().combine_user_item_representations(u_repr, i_embed)

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return ItemFeatureMatrixFactorization(dataset, conf['feature_name'], conf['aggregate_for_rec'], conf['lambda_content'],
                                              conf['temperature'], conf['embedding_loss_aggregator'], conf['intermediate_layers'],
                                              conf['embedding_dim'], conf['use_user_bias'], conf['use_item_bias'],
                                              conf['use_global_bias'])


class UserFeatureMatrixFactorization(_FeatureMatrixFactorization):
    """algorithms/sgd_alg.py:1508-1614. The reference feeds [B, 1, D] to InfoNCE ("blow up first dimension"): every group holds
    one pair, so the contrastive loss is identically zero — reproduced as is."""
    _side = 'user'

    def forward(self, u_idxs, i_idxs):
        u_repr = self.get_user_representations(u_idxs)
        i_repr = self.get_item_representations(i_idxs)
        dots = self.combine_user_item_representations(u_repr, i_repr)
        self.compute_reg_losses(u_repr[0][:, None, :], u_repr[1][:, None, :])
        return dots

    def get_user_representations(self, u_idxs):
        profile = ops.LookupFn.apply(self.user_embeddings.weight, u_idxs)
        content = self.embedding_net(u_idxs).squeeze(dim=1)
        if self.use_user_bias:
            return profile, content, ops.LookupFn.apply(self.user_bias.weight, u_idxs).squeeze()
        return profile, content

    def combine_user_item_representations(self, u_repr, i_repr):
        u_embed = self._mean2(u_repr[0], u_repr[1]) if self.aggregate_for_rec else u_repr[0]
        if self.use_user_bias:
            return super().combine_user_item_representations((u_embed, u_repr[-1]), i_repr)
        return super().combine_user_item_representations(u_embed, i_repr)

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return UserFeatureMatrixFactorization(dataset, conf['feature_name'], conf['aggregate_for_rec'], conf['lambda_content'],
                                              conf['temperature'], conf['embedding_loss_aggregator'], conf['intermediate_layers'],
                                              conf['embedding_dim'], conf['use_user_bias'], conf['use_item_bias'],
                                              conf['use_global_bias'])
