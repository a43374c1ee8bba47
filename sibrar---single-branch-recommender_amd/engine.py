"""Fused training step: the whole hot loop body of train/trainer.py:204-223 as ONE straight-line sequence of HIP kernel
launches — forward, losses, hand-written backward, gradient all-reduce, fused dense optimizer — without the autograd
tape, without per-op tensor allocation and without a host sync.

It computes exactly what ``loss(model(u, i)).backward(); optimizer.step()`` computes through the ``nn.Module`` surface
(``tests/test_hip_golden.py::test_fused_step_matches_autograd_path`` and ``..._on_config_variants`` check losses and
parameters after several steps): the modules keep owning the parameters (flat buffer of ``optim.FlatParameters``), this class
only replaces the launch choreography.

Host work per step is split in two: ``prepare`` (modality draw, per-modality counts, one packed upload — normally on the
loader thread, ahead of time) and ``step`` (wait for the upload's event, copy it into the captured graph's static buffers,
replay forward + backward as a hipGraph, gradient exchange, one fused optimizer launch). Signatures that have not been
captured yet run the same launches one by one (~120 ctypes calls).

Supported (everything the shipped sbnet configurations use): user side = embedding lookup / any plain FeatureEmbedding or a
SingleBranchNetEntity; item side = SingleBranchNetEntity; rec losses bce / bpr / sampled softmax; InfoNCE regularisation.
"""
from __future__ import annotations

import atexit
import ctypes
import math
import os
import weakref
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import ops, parallel
from ._lib import call, pin_stream, ptr, to_device
from .config import EmbeddingRegularizationType
from .losses import RecBayesianPersonalizedRankingLoss, RecBinaryCrossEntropy, RecSampledSoftmaxLoss
from .sbnet import FeatureEmbedding, SingleBranchNet, SingleBranchNetEntity, resolve_rows


class Arena:
    """Bump allocator over one device buffer, reset at the start of every step (no torch.empty per activation)."""

    def __init__(self, device):
        self.device = device
        self.buf = torch.empty(1 << 22, device=device, dtype=torch.uint8)
        self.off = 0
        self.high = 0

    def reset(self):
        if self.high > self.buf.numel():
            self.buf = torch.empty(int(self.high * 1.25), device=self.device, dtype=torch.uint8)
        self.off = 0
        self.high = 0

    def _take(self, nbytes, dtype, shape):
        start = (self.off + 255) & ~255
        self.off = start + nbytes
        self.high = max(self.high, self.off)
        if self.off > self.buf.numel():
            return torch.empty(shape, device=self.device, dtype=dtype)      # overflow: plain allocation this step, grow next
        return self.buf[start:start + nbytes].view(dtype).view(shape)

    def f32(self, *shape):
        n = int(np.prod(shape)) if shape else 1
        return self._take(4 * n, torch.float32, shape)

    def i32(self, *shape):
        n = int(np.prod(shape)) if shape else 1
        return self._take(4 * n, torch.int32, shape)

    def f64(self, *shape):
        n = int(np.prod(shape)) if shape else 1
        return self._take(8 * n, torch.float64, shape)

    def u8(self, *shape):
        n = int(np.prod(shape)) if shape else 1
        return self._take(n, torch.uint8, shape)


def _grad_of(p: torch.Tensor) -> torch.Tensor:
    if p.grad is None:
        raise RuntimeError('FusedTrainStep needs optim.FlatParameters gradient views (build the FusedOptimizer first)')
    return p.grad


class _EntityRun:
    """Forward state + backward of one SingleBranchNetEntity call."""

    def __init__(self, ent: SingleBranchNetEntity, arena: Arena):
        self.ent, self.a = ent, arena
        cfg = ent.entity_config
        self.C, self.D = cfg.common_modality_dim, ent.output_dim
        self.layers = ent.sb_net[ent._poly_index].layer_plan()
        self.trailing = ent.sb_net[ent._poly_index + 1] if ent._trailing_bn else None
        self.p_drop = cfg.single_branch_input_dropout
        self.normalize = cfg.normalize_single_branch_input
        self.reg = ent._reg_type != EmbeddingRegularizationType.NoRegularization
        self.tau, self.reg_w = float(cfg.regularization_temperature), float(cfg.regularization_weight)
        self._ws = {}
        self._cs_ws = {}             # id(linear) -> column-reduction workspace of its folded bias gradient
        # set by FusedTrainStep for the ITEM side: the trailing BatchNorm is applied inside the scorer (csrc/fused_tail.hip)
        # whenever a step draws one modality per slot; the normalised representation is then never stored
        self.fuse_tail = False
        self.tail = None             # (z, mean, rstd) of the current step when the tail is fused
        self.fold = True             # bias-gradient column sums ride on the kernels that produce their input
        self.tn = ops.DeferredTN()                       # dW slabs summed by one launch
        self._dx0 = {}               # (R, C) -> persistent [R + 1, C] slot-gradient buffer with a zero sentinel row (graph mode)

    # ---- forward -----------------------------------------------------------------------------------------------------
    def plan(self, draw: Tuple[np.ndarray, list], pad: bool = False):
        """Host part of a step: -> (modality position of every slot int8 [R], k, rows per modality, order, R, padded).
        The stable counting sort itself (slot lists per modality) runs on the GPU (``sbr_partition_slots``); the host only
        counts, because the per-modality row counts are the launch sizes.

        ``pad`` (graph mode): the number of rows a modality gets is a random variable (one or two modalities are drawn per
        index, sgd_alg.py:1912-1927), but a captured graph has fixed launch sizes. Each modality's slot list is therefore
        padded to the next multiple of a bucket (>= 2 sigma of the binomial count, so a few signatures cover all steps)
        with the sentinel slot R: padded launches read entity idx[R / k] (a copy of idx[0] kept behind the index buffer),
        write row R of the [R + 1, C] modality matrix — which the shared network never reads — and see a zero gradient row,
        so they add exact zeros to every parameter gradient."""
        pos, order = draw
        flat = np.ascontiguousarray(pos.reshape(-1), dtype=np.int8)
        R = flat.size
        if len(order) > ops.PARTITION_MAX_MODALITIES:
            raise NotImplementedError(f'{len(order)} modalities on one entity (the slot partition kernel handles '
                                      f'{ops.PARTITION_MAX_MODALITIES})')
        # per-modality counts: one vectorised compare-and-count per modality (np.bincount converts the int8 array to intp first:
        # 0.1 ms at 90k slots, a quarter of the whole prepare stage)
        counts = np.array([np.count_nonzero(flat == m) for m in range(len(order))], dtype=np.int64)
        if pad:
            # half a bucket on either side of the expected count must cover the draw's spread for practically every step: a
            # count outside means a new signature — a step of plain launches and, at its second sighting, a capture (~12 ms:
            # with 3.4 sigma per side one bench run in eight measured 0.98 instead of 0.86 ms per step). sigma <= sqrt(R) / 2, so
            # a bucket of 5 sqrt(R) (rounded up to 64) is >= 5 sigma per side (c2 at B = 8192: 1536, i.e. 768 padded rows
            # on average instead of 512).
            bucket = max(64, -(-int(5.0 * math.sqrt(R)) // 64) * 64)
            # bucket grid shifted so that the expected count R / n_modalities sits in the MIDDLE of a bucket (+- 2 sigma and
            # more on either side): one signature then covers nearly every step. With the grid at multiples of the bucket a
            # count whose mean is such a multiple (c2: 45056 = 44 * 1024) flips between two capacities from step to step.
            off = (R // len(order) + bucket // 2) % bucket
            counts = np.where(counts > 0, (np.maximum(counts - off, 0) + bucket - 1) // bucket * bucket + off, 0)
        return flat, pos.shape[1], tuple(int(c) for c in counts), tuple(order), R, bool(pad)

    def forward(self, idx: torch.Tensor, plan, seed: int, pos_dev: Optional[torch.Tensor] = None):
        """Launches only (graph-capturable) when ``pos_dev`` — the device copy of plan[0] — is handed in."""
        ent, a, st = self.ent, self.a, ops.stream()
        pos_flat, k, counts, order, R, padded = plan
        self.padded = padded
        dev = idx.device
        if pos_dev is None:
            pos_dev = to_device(torch.from_numpy(pos_flat), dev)
        seg = [0]
        for c in counts:
            seg.append(seg[-1] + c)
        slots = a.i32(seg[-1])
        seg_arr = (ctypes.c_int * len(seg))(*seg)
        ws = a.i32(int(ops.lib().sbr_partition_slots_workspace(R)) // 4 + 8)
        call('sbr_partition_slots', ptr(pos_dev), R, len(counts), ctypes.cast(seg_arr, ctypes.c_void_p), ptr(slots), ptr(ws),
             ws.numel() * 4, st)
        entries, tables, offs = [], [], [0]
        for m, c in enumerate(counts):
            if c:
                fe = ent.modality_modules[order[m]]
                entries.append((fe, offs[-1], c))
                tables.append(fe._table)
                offs.append(offs[-1] + c)
        idx_flat = idx.reshape(-1)
        rows, _ = resolve_rows(idx_flat, k, slots, offs, tables, ent._idx_err)
        self.entries, self.rows, self.slots, self.R, self.k, self.shape = entries, rows, slots, R, k, tuple(idx.shape)
        x0 = a.f32(R + 1 if padded else R, self.C)      # row R: landing row of the padded launches
        self.hidden = [fe.front_forward(fe.front_params(), rows[o:o + n], n, x0, slots[o:o + n]) for fe, o, n in entries]
        self.x0 = x0
        x = x0[:R]
        if self.normalize:
            xn, self.inv = a.f32(R, self.C), a.f32(R)
            call('sbr_l2norm_fwd', ptr(x), ptr(xn), ptr(self.inv), R, self.C, ops.NORM_EPS, st)
            self.xn = x = xn
        self.seed = None
        if self.p_drop:
            xd = a.f32(R, self.C)
            # seed = (step seed in device memory, part of the batch upload) + a per-side offset: replayable in a hipGraph
            call('sbr_dropout_dev', ptr(x), ptr(xd), x.numel(), float(self.p_drop), ptr(seed[0]), int(seed[1]), st)
            self.seed, x = seed, xd
        self.acts = []                                   # per layer: (input, pre-BN output | None, output, mean, rstd)
        # statistics of the trailing BatchNorm from the epilogue of the GEMM in front of it (fused tail only: nothing but the
        # statistics is needed from that pass over the output)
        tail_stats = (self.trailing is not None and self.fuse_tail and k == 1 and not self.reg and bool(self.layers)
                      and self.layers[-1][1] is None and ops.lib().sbr_bn_score_supported(int(self.layers[-1][0].weight.shape[0])))
        self._stats_folded = False
        for li, (lin, bn, act) in enumerate(self.layers):
            w = lin.weight
            if bn is None:
                out_l = a.f32(R, w.shape[0])
                if tail_stats and li == len(self.layers) - 1 and ops.linear_nt_stats_ok(x, w, out_l):
                    # ... and the BatchNorm's finalisation (batch mean / rstd, running statistics) by the GEMM's last workgroup
                    bn = self.trailing
                    self._fin = (a.f32(w.shape[0]), a.f32(w.shape[0]))
                    y = ops.linear_nt(x, w, lin.bias, act, out=out_l, stats_ws=self._bn_ws(bn, w.shape[0]),
                                      bn_fin=(self._bn_arrive(bn), bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                              self._fin[0], self._fin[1], ops.BN_EPS, ops.BN_MOMENTUM))
                    self._stats_folded = True
                else:
                    y = ops.linear_nt(x, w, lin.bias, act, out=out_l)
                self.acts.append((x, None, y, None, None))
            else:
                z = ops.linear_nt(x, w, lin.bias, 0, out=a.f32(R, w.shape[0]))
                y, mean, rstd = self._bn_fwd(bn, z, act)
                self.acts.append((x, z, y, mean, rstd))
            x = y
        self.tb = None
        self.tail = None
        if self.trailing is not None:
            if self.fuse_tail and k == 1 and not self.reg and ops.lib().sbr_bn_score_supported(int(x.shape[1])):
                # statistics only: the scorer normalises on the fly (FusedTrainStep._phase1), nothing else reads the output
                bn, n_, D_ = self.trailing, x.shape[0], x.shape[1]
                if self._stats_folded:
                    mean, rstd = self._fin                 # written by the last workgroup of the GEMM in front (no launch here)
                else:
                    mean, rstd = a.f32(D_), a.f32(D_)
                    call('sbr_bn_train_stats', ptr(x), n_, D_, ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked),
                         ptr(mean), ptr(rstd), ptr(self._bn_ws(bn, D_)), ops.BN_EPS, ops.BN_MOMENTUM, st)
                self.tail = (x, mean, rstd)
                self.e = None
                self.reg_loss = None
                return None
            y, mean, rstd = self._bn_fwd(self.trailing, x, 0)
            self.tb = (x, y, mean, rstd)
            x = y
        self.e = x                                       # [R, D] == [S, k, D]
        S = R // k
        self.reg_loss = None
        if self.reg:
            if k != 2:
                raise SystemError('second last dimension of embeddings should be of size 2')
            N = self.shape[-1]
            G = S // N
            self.G, self.N = G, N
            self.reg_loss = a.f64()
            e3 = x.view(S, 2, self.D)
            ops.infonce_fwd(e3[:, 0].data_ptr(), e3[:, 1].data_ptr(), 2 * self.D, G, N, self.D, self.tau, 1.0 / (G * N),
                            self.reg_loss, x.device)
        if k == 1:
            return x.view(S, self.D)
        out = a.f32(S, self.D)
        self.arg = a.u8(S, self.D) if ent._agg_mode == 1 else None
        call('sbr_aggregate_fwd', ptr(x), ptr(out), ptr(self.arg), S, k, self.D, ent._agg_mode, st)
        return out

    def _bn_arrive(self, bn):
        """Persistent zeroed int64[1] of one BatchNorm: the arrival counter of the GEMM that finalises its statistics."""
        c = self._ws.get(('arrive', id(bn)))
        if c is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('the BatchNorm arrival counter must exist before a step is captured (run one plain step first)')
            c = self._ws[('arrive', id(bn))] = torch.zeros(1, device=bn.weight.device, dtype=torch.int64)
        return c

    def _bn_ws(self, bn, D):
        """Persistent column-reduction workspace of one BatchNorm (zero on first use, left zeroed by every kernel pair)."""
        ws = self._ws.get(id(bn))
        if ws is None:
            ws = self._ws[id(bn)] = torch.zeros(ops.COLRED_WS_FACTOR * 2 * D, device=bn.weight.device, dtype=torch.float64)
        return ws

    def _bn_fwd(self, bn, x, act):
        a, st = self.a, ops.stream()
        n, D = x.shape
        y, mean, rstd, ws = a.f32(n, D), a.f32(D), a.f32(D), self._bn_ws(bn, D)
        call('sbr_bn_train_fwd', ptr(x), ptr(y), n, D, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
             ptr(bn.num_batches_tracked), ptr(mean), ptr(rstd), ptr(ws), ops.BN_EPS, ops.BN_MOMENTUM, act, st)
        return y, mean, rstd

    def _bn_bwd(self, bn, dy, y, x, mean, rstd, act):
        a, st = self.a, ops.stream()
        n, D = x.shape
        dx, ws = a.f32(n, D), self._bn_ws(bn, D)
        call('sbr_bn_train_bwd', ptr(dy), ptr(y), ptr(x), ptr(dx), n, D, ptr(bn.weight), ptr(mean), ptr(rstd),
             ptr(_grad_of(bn.weight)), ptr(_grad_of(bn.bias)), ptr(ws), act, st)
        return dx

    # ---- backward ------------------------------------------------------------------------------------------------------
    def _fold_ws(self, lin, C):
        ws = self._cs_ws.get(id(lin))
        if ws is None:
            ws = self._cs_ws[id(lin)] = ops.new_colsum_ws(lin.weight.device, C)
        return ws

    def backward(self, dout: torch.Tensor, one_f32: torch.Tensor, tail=None):
        """``tail`` = (dlogits [B, N], user representations [B, D]) when the trailing BatchNorm was fused into the scorer: the
        gradient of the normalised representation is dlogits[s] * u[b(s), :] and is never stored."""
        ent, a, st = self.ent, self.a, ops.stream()
        R, k, D = self.R, self.k, self.D
        S = R // k
        pending = [] if self.fold else None           # folded bias-gradient column sums, completed by ONE launch at the end
        last_lin = self.layers[-1] if self.layers else None
        folded_last = False
        if tail is not None:
            dlog, ur = tail
            z, mean, rstd = self.tail
            bn = self.trailing
            d = a.f32(R, D)
            ws_col = None
            if pending is not None and last_lin is not None and last_lin[1] is None and not last_lin[2] \
                    and last_lin[0].bias is not None and ops.colsum_supported(D):
                # the Linear in front of the BatchNorm has no BatchNorm / activation of its own: d IS its pre-activation
                # gradient, whose column sums (the bias gradient) the apply pass accumulates on the way
                ws_col = self._fold_ws(last_lin[0], D)
                pending.append((ws_col, _grad_of(last_lin[0].bias)))
                folded_last = True
            Bq = dlog.shape[0]
            call('sbr_bn_score_bwd_apply', ptr(dlog), ptr(ur), ptr(z), ptr(d), Bq, R // Bq, D, ptr(bn.weight), ptr(mean), ptr(rstd),
                 ptr(self._bn_ws(bn, D)), ptr(_grad_of(bn.weight)), ptr(_grad_of(bn.bias)), ptr(ws_col), st)
        else:
            if k == 1:
                de = dout
            else:
                de = a.f32(R, D)
                call('sbr_aggregate_bwd', ptr(dout), ptr(self.arg), ptr(de), S, k, D, ent._agg_mode, st)
            if self.reg:
                dreg = a.f32(R, D)
                e3, d3 = self.e.view(S, 2, D), dreg.view(S, 2, D)
                # d(total)/d(reg_loss) = regularization_weight (sgd_alg.py:2002); mean over G*N rows inside the kernel
                ops.infonce_bwd(e3[:, 0].data_ptr(), e3[:, 1].data_ptr(), 2 * D, self.G, self.N, D, self.tau,
                                self.reg_w / (self.G * self.N), one_f32, d3[:, 0].data_ptr(), d3[:, 1].data_ptr(), 2 * D, dreg.device)
                de = de.add_(dreg) if k > 1 else dreg.add_(de)
            d = de
            if self.tb is not None:
                x, y, mean, rstd = self.tb
                d = self._bn_bwd(self.trailing, d, y, x, mean, rstd, 0)
        # gradient of the [R (+1), C] modality matrix: the last producer below writes rows [0, R); the sentinel row is zero
        if self.padded:
            # [R + 1, C] with a zero sentinel row: a persistent buffer per (R, C), zeroed when it is created — nothing below writes
            # row R, so it needs no fill launch per step (4.5 us for 512 bytes). Like the arena it never moves, so captured steps
            # keep a valid address.
            dx0 = self._dx0.get((R, self.C))
            if dx0 is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError('the slot-gradient buffer must exist before a step is captured (run one plain step first)')
                dx0 = self._dx0[(R, self.C)] = torch.zeros(R + 1, self.C, device=a.device, dtype=torch.float32)
        else:
            dx0 = a.f32(R, self.C)
        tail_ops = (self.seed is not None) + bool(self.normalize)
        chain = list(zip(reversed(self.layers), reversed(self.acts)))
        pre = None                                     # dz of the coming layer when the previous NN product already produced it
        for li, ((lin, bn, act), (x, z, y, mean, rstd)) in enumerate(chain):
            w = lin.weight
            bias_done = folded_last and li == 0
            if pre is not None:
                dz, pre, bias_done = pre, None, True
            elif bn is not None:
                dz = self._bn_bwd(bn, d, y, z, mean, rstd, act)
            elif act and pending is not None and ops.colsum_supported(d.shape[1]):
                ws_l = self._fold_ws(lin, d.shape[1])
                dz = a.f32(d.shape[0], d.shape[1])
                call('sbr_act_grad_gather_colsum', ptr(d), ptr(y), d.stride(0), None, ptr(dz), dz.stride(0), d.shape[0], d.shape[1],
                     act, ptr(ws_l), st)
                pending.append((ws_l, _grad_of(lin.bias)))
                bias_done = True
            else:
                dz = ops.act_grad(d, y, act) if act else d
            self.tn.matmul_tn(id(lin), dz, x, out=_grad_of(w))
            if not bias_done:
                ops.colsum(dz, out=_grad_of(lin.bias))
            last = li == len(self.layers) - 1
            if not last:
                # the layer in front (x is its output): Linear + activation without a BatchNorm of its own -> the NN product
                # applies the activation derivative and accumulates that layer's bias gradient in its epilogue
                lin_b, bn_b, act_b = chain[li + 1][0]
                nxt = a.f32(R, w.shape[1])
                if bn_b is None and act_b and pending is not None and lin_b.bias is not None \
                        and ops.matmul_nn_actgrad_ok(dz, w, x, nxt) and ops.colsum_supported(w.shape[1]):
                    ws_b = self._fold_ws(lin_b, w.shape[1])
                    pre = ops.matmul_nn_actgrad(dz, w, x, act_b, nxt, ws_b)
                    pending.append((ws_b, _grad_of(lin_b.bias)))
                    d = None
                    continue
                d = ops.matmul_nn(dz, w, out=nxt)
            else:
                d = ops.matmul_nn(dz, w, out=dx0[:R] if not tail_ops else a.f32(R, w.shape[1]))
        if self.seed is not None:
            tail_ops -= 1
            dd = dx0[:R] if not tail_ops else a.f32(R, self.C)
            call('sbr_dropout_dev', ptr(d), ptr(dd), d.numel(), float(self.p_drop), ptr(self.seed[0]), int(self.seed[1]), st)
            d = dd
        if self.normalize:
            dn = dx0[:R]
            call('sbr_l2norm_bwd', ptr(d), ptr(self.xn), ptr(self.inv), ptr(dn), R, self.C, ops.NORM_EPS, st)
            d = dn
        if d.data_ptr() != dx0.data_ptr():               # no layer at all: the incoming gradient is the matrix gradient
            torch.mul(d, 1.0, out=dx0[:R])                # a kernel node, not a memcpy node (see DESIGN.md §5 on memset nodes)
        d = dx0
        for (fe, o, n), hs in zip(self.entries, self.hidden):
            ps = fe.front_params()
            fe.front_backward(ps, hs, self.rows[o:o + n], n, self.x0, d, self.slots[o:o + n], grad_out=[_grad_of(p) for p in ps],
                              pending=pending, tn=self.tn)
        took = self.tn.finish(pending)                 # ... and the pending bias-gradient column sums, in the same launch
        if pending and not took:
            ops.colred_finish(pending)


_IDENTITY_SLOTS = {}


def _identity_slots(n: int, device) -> torch.Tensor:
    """0 .. n-1 as int32 on ``device``, built once per (n, device): a read-only constant of every step (a torch.arange per step is a
    5 us launch inside the captured step)."""
    key = (n, str(device))
    t = _IDENTITY_SLOTS.get(key)
    if t is None:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return torch.arange(n, device=device, dtype=torch.int32)       # never allocate a cached constant inside a capture
        t = _IDENTITY_SLOTS[key] = torch.arange(n, device=device, dtype=torch.int32)
    return t


class _PlainRun:
    """Forward / backward of a plain FeatureEmbedding side (embedding lookup or projector on one feature)."""

    def __init__(self, fe: FeatureEmbedding, arena: Arena):
        if fe.post_embedding_layers is not None:
            raise NotImplementedError('post_embedding_layers are not part of the SingleBranchNet path')
        self.fe, self.a = fe, arena
        self.reg_loss = None
        self.xchg = None             # (grad rows [cap, D], table rows int32 [cap]) of a data-parallel sparse exchange

    def plan(self, draw=None, pad=False):
        return None

    def forward(self, idx, plan=None, seed=0, slots=None):
        fe, a = self.fe, self.a
        flat = idx.reshape(-1)
        n = flat.numel()
        self.n = n
        self.slots = _identity_slots(n, flat.device)
        if fe.kind == 'categorical' and flat.dtype == torch.int64 and flat.is_contiguous():
            # embedding lookup: id -> row and the row gather in one launch
            W = fe.front_params()[0]
            self.out = a.f32(n, fe.front_dim)
            if ops.lib().sbr_lookup_rows_supported(ptr(W), W.stride(0), ptr(self.out), self.out.stride(0), int(fe.front_dim)):
                t = fe._table
                self.rows = a.i32(n)
                call('sbr_lookup_rows', ptr(flat), n, ptr(t.rowmap), int(t.rowmap.numel()) if t.rowmap is not None else int(t.n_rows),
                     ptr(W), W.stride(0), ptr(self.rows), ptr(self.out), self.out.stride(0), int(fe.front_dim), ptr(fe._idx_err),
                     ops.stream())
                self.hidden = []
                return self.out
        self.rows, _ = resolve_rows(flat, 1, self.slots, [0, n], [fe._table], fe._idx_err)
        self.out = a.f32(n, fe.front_dim)
        self.hidden = fe.front_forward(fe.front_params(), self.rows, n, self.out, None)
        return self.out

    def backward(self, dout, one_f32):
        fe = self.fe
        ps = fe.front_params()
        if self.xchg is not None:
            # data-parallel run, lookup table: the rows of this batch (table row, gradient row) go into the send buffer of the
            # all-gather instead of the dense table gradient; FusedTrainStep scatters every rank's rows after the exchange
            # (_scatter_user_rows). Unused capacity: table row 0 with a zero gradient row.
            g, r = self.xchg
            n = self.n
            if n > g.shape[0]:
                raise RuntimeError(f'batch of {n} rows after the exchange capacity was fixed at {g.shape[0]} (the first batch '
                                   f'of a data-parallel run must be a full one)')
            torch.mul(dout, 1.0, out=g[:n])               # kernel nodes (no memcpy / memset nodes in the captured step)
            torch.add(self.rows, 0, out=r[:n])
            if n < g.shape[0]:
                g[n:].fill_(0.0)
                r[n:].fill_(0)
            return
        fe.front_backward(ps, self.hidden, self.rows, self.n, self.out, dout, None, grad_out=[_grad_of(p) for p in ps])


class DeferredTable:
    """Deferred row-wise Adam / AdamW for ONE lookup table (``sbr_adam_rows`` / ``sbr_adam_step_rows``; csrc/optim.hip explains why
    the result is bit-identical to the dense optimizer of train/trainer.py:62-68). ``catch_up`` before the forward pass of a step,
    ``step`` = the step's whole optimizer launch (row updates of this table + the dense step of every other parameter),
    ``flush`` before anybody else reads the table: the model's state_dict, its module-level forward (evaluation, autograd path), a
    dense optimizer step, ``FusedTrainStep.close``."""

    # Sweep period W of the optimizer launch: every row is brought up to date at least every W steps (0: no sweep; None: chosen per
    # table — 16, or more when 1 / 16 of the table per step would dwarf the dense part of the launch: the sweep's bytes stay within
    # a quarter of the dense part's (at least 8 MB), W <= 1024. c2: 16 (19 MB per step beside 185 MB); c4's 1M x 256 table: 768).
    SWEEP_EVERY = None

    def __init__(self, opt, param: torch.Tensor, lo: int, hi: int, rowmap: Optional[torch.Tensor]):
        n_rows, D = param.shape
        assert param.stride() == (D, 1) and hi - lo == n_rows * D
        self.opt, self.lo, self.hi, self.n_rows, self.D, self.rowmap = opt, lo, hi, n_rows, D, rowmap
        fp = opt.fp
        self.p, self.g = fp.flat[lo:hi], fp.grad[lo:hi]
        self.m, self.v = opt.m[lo:hi], opt.v[lo:hi]
        dev = self.p.device
        n_sub = (D + 63) // 64                   # one bookkeeping entry per 64-element sub-row (the work of one wave)
        self.last = torch.zeros(n_rows * n_sub, device=dev, dtype=torch.int32)
        self.claim = torch.zeros(n_rows * n_sub, device=dev, dtype=torch.int32)
        self.sched = torch.zeros(4096, 2, device=dev, dtype=torch.float32)
        # the sweep of the optimizer launch (csrc/optim.hip, adam_step_rows_kernel): 1 / SWEEP_EVERY of the table's sub-rows per step
        self.n_q = n_rows * n_sub
        self._cursor = 0
        self._caught = -1            # step whose catch-up has run (its claims are how the sweep tells that step's rows)
        self.sweep_ok = True         # False: the optimizer's ids are not the catch-up's (sparse data-parallel exchange)
        self.kind = 0 if opt.name == 'adamw' else 1
        self.flushed_to = 0          # step up to which EVERY row is known to be current
        if opt.step_count > 0:       # steps taken densely before this object existed: every row is current
            self.last.fill_(opt.step_count)
            self.flushed_to = opt.step_count

    def sweep_period(self) -> int:
        if self.SWEEP_EVERY is not None:
            return int(self.SWEEP_EVERY)
        table = self.hi - self.lo
        dense_bytes = 28.0 * (self.opt.fp.total - table)
        return int(min(1024, max(16, -(-24.0 * table // max(0.25 * dense_bytes, 8e6)))))

    def _grow_sched(self, step: int):
        if step >= self.sched.shape[0]:
            bigger = torch.zeros(2 * max(step, self.sched.shape[0]), 2, device=self.sched.device, dtype=torch.float32)
            bigger[:self.sched.shape[0]] = self.sched
            self.sched = bigger

    @staticmethod
    def _ids(ids):
        ids64 = ids if ids is not None and ids.dtype == torch.int64 else None
        ids32 = ids if ids is not None and ids.dtype == torch.int32 else None
        return ids64, ids32

    def _call(self, mode: int, ids, step: int):
        self._grow_sched(step)
        ids64, ids32 = self._ids(ids)
        o = self.opt
        call('sbr_adam_rows', self.kind, mode, ptr(self.p), ptr(self.g), ptr(self.m), ptr(self.v), self.n_rows, self.D, ptr(ids64),
             ptr(ids32), ptr(self.rowmap) if ids64 is not None else None, 0 if ids is None else ids.numel(), ptr(self.claim),
             ptr(self.last), ptr(self.sched), float(o.lr), 0.9, 0.999, 1e-8, float(o.wd), int(step), ops.stream())

    def catch_up(self, ids: torch.Tensor):
        """ids: the entity ids (int64, mapped through the table's id map) or table rows (int32) the coming step reads."""
        self._call(0, ids, self.opt.step_count + 1)
        self._caught = self.opt.step_count + 1

    def update(self, ids: torch.Tensor):
        """Apply step ``opt.step_count`` (already counted) to the rows that received gradient; zeroes those gradient rows."""
        self._call(1, ids, self.opt.step_count)

    def step(self, ids: torch.Tensor, copy=None) -> bool:
        """The optimizer launch of a step (optimizer.step() + zero_grad() of train/trainer.py:221-222): this table's rows named by
        ``ids`` take the step with their gradient rows, every other parameter of the flat buffer takes the dense step, consumed
        gradient elements are reset, ``copy`` = (src, dst) float64 tensors ride along. Returns True when ``copy`` was made."""
        o = self.opt
        o.step_count += 1
        self._grow_sched(o.step_count)
        ids64, ids32 = self._ids(ids)
        src, dst = copy if copy is not None else (None, None)
        fp = o.fp
        # the sweep needs the claims of this step's catch-up (same ids) to tell the batch's rows from the others
        n_sweep = 0
        every = self.sweep_period()
        if every > 0 and self.sweep_ok and self._caught == o.step_count:
            n_sweep = -(-self.n_q // every)
        sweep_lo = self._cursor
        self._cursor = (self._cursor + n_sweep) % self.n_q
        call('sbr_adam_step_rows', self.kind, ptr(fp.flat), ptr(fp.grad), ptr(o.m), ptr(o.v), fp.total, self.lo, self.hi, self.D, ptr(ids64),
             ptr(ids32), ptr(self.rowmap) if ids64 is not None else None, 0 if ids is None else ids.numel(), ptr(self.claim),
             ptr(self.last), ptr(self.sched), float(o.lr), 0.9, 0.999, 1e-8, float(o.wd), int(o.step_count), sweep_lo, n_sweep,
             ptr(src), ptr(dst), 0 if src is None else src.numel(), ops.stream())
        return src is not None

    def flush(self):
        t = self.opt.step_count
        if t > self.flushed_to:
            self._call(2, None, t)
            self.flushed_to = t

    def mark_all_current(self):
        """A dense optimizer step has just updated every row (FusedOptimizer.step_flat without ``skip``)."""
        self.last.fill_(self.opt.step_count)
        self.flushed_to = self.opt.step_count


class FusedTrainStep:
    """``use_graph`` (default: on, env ``SBR_GRAPH=0`` turns it off): forward + backward of a step are captured once per
    batch signature into a hipGraph and replayed — one launch instead of ~50. The step is made a pure function of device
    buffers: the modality plans are padded to bucketed capacities (``_EntityRun.plan``), the dropout seed of the step travels
    in the batch upload and is read from device memory by the dropout kernels (``sbr_dropout_dev``).
    The gradient all-reduce and the optimizer launch stay outside the graph (RCCL call, step-dependent scalars)."""

    MAX_GRAPHS = 16

    def __init__(self, net: SingleBranchNet, rec_loss, optimizer, use_graph: Optional[bool] = None):
        if not isinstance(net.item_embedding_module, SingleBranchNetEntity):
            raise NotImplementedError('FusedTrainStep needs a SingleBranchNetEntity item side')
        self.net, self.rec_loss, self.opt = net, rec_loss, optimizer
        dev = net.device
        self.arena = Arena(dev)
        self.user = (_EntityRun if net.is_user_sb_module else _PlainRun)(net.user_embedding_module, self.arena)
        self.item = _EntityRun(net.item_embedding_module, self.arena)
        self.item.fuse_tail = True   # the trailing BatchNorm runs inside the scorer whenever a step draws one modality per slot
        self.kind = {RecBinaryCrossEntropy: 0, RecBayesianPersonalizedRankingLoss: 1, RecSampledSoftmaxLoss: 2}[type(rec_loss)]
        self.one64 = torch.ones((), device=dev, dtype=torch.float64)
        self.one32 = torch.ones((), device=dev, dtype=torch.float32)
        self.n_steps = 0
        if use_graph is None:
            use_graph = os.environ.get('SBR_GRAPH', '1') != '0'
        self.use_graph = bool(use_graph)
        self._n_prepared = 0
        self._graphs = {}            # batch signature -> None (seen once, run eagerly) | _CapturedStep
        self._arena_buf = None
        self.n_replays = 0
        self._up_stream = None
        self._label_cache = {}
        self.last_out3 = None        # the last step's (total, rec, reg) as ONE float64 [3] tensor (graph replays; else None)
        self._loss_ws = None         # partial sums + arrival counter of the one-launch loss kernel (zeroed once, self-resetting)
        self._tail_ws = None         # the same for the fused scorer + loss + statistics kernel
        self._packed = None
        # two-phase launch (and two graphs) when gradients are exchanged: see _reduce_user_part. SBR_FORCE_SPLIT=1 exercises
        # the same launch structure on one GPU.
        self.split = parallel.is_distributed() or os.environ.get('SBR_FORCE_SPLIT', '0') == '1'
        self._urange = self._user_range()
        # sparse exchange of a lookup user side: a batch touches at most B rows of the [U, D] table, so the ranks all-gather
        # (row index, gradient row) pairs instead of all-reducing the dense table gradient (c2 at 8 GPUs moves
        # 8 x 4.2 MB instead of ring-reducing 51 MB). Decided at the first step (_setup_sparse_exchange), SBR_SPARSE_EXCHANGE=0
        # keeps the dense all-reduce.
        self._sparse = None          # None: undecided | False: dense | (send, recv, cap)
        if not (parallel.is_distributed() and isinstance(self.user, _PlainRun) and self.user.fe.kind == 'categorical'
                and self._urange is not None and os.environ.get('SBR_SPARSE_EXCHANGE', '1') != '0'):
            self._sparse = False
        _LIVE.add(self)
        self.opt.zero_grad()
        # lookup user table updated row by row instead of densely (DeferredTable). Needs to know which rows received
        # gradient: the batch's users on one GPU, the all-gathered row lists of the sparse exchange in a data-parallel run.
        # Bit-identical to the dense optimizer (tests/test_hip_kernels.py::test_deferred_row_wise_adam_replay_is_bit_identical);
        # on by default (SBR_DEFERRED_ADAM=0: dense): the untouched rows of the table — 92 % of c2's user table at B = 8192, 99.7 %
        # at the reference's batch 256 — leave the optimizer's HBM stream (c2: 358 of 541 MB per step). One extra launch per step
        # (the catch-up of the rows a batch reads); the update rides on the optimizer launch (sbr_adam_step_rows).
        self.deferred = None
        fe = net.user_embedding_module
        if (isinstance(self.user, _PlainRun) and fe.kind == 'categorical' and self._urange is not None
                and optimizer.name in ('adamw', 'adam') and optimizer.deferred is None
                and os.environ.get('SBR_DEFERRED_ADAM', '1') != '0'
                and (not parallel.is_distributed() or self._sparse is not False)):
            table = fe.front_params()[0]
            lo, hi = self._urange
            if table.dim() == 2 and table.stride() == (table.shape[1], 1) and hi - lo == table.numel():
                self.deferred = DeferredTable(optimizer, table, lo, hi, fe._table.rowmap)
                optimizer.deferred = self.deferred
                # Readers AND writers of the table outside the fused step see / leave a current table: state_dict() and the
                # module-level forward flush first; load_state_dict() (Trainer / load_model_from_path, on the net or on the user
                # module alone) flushes BEFORE the copy — the zero-gradient steps a row still owes belong to the OLD weights and
                # moments, replayed after the copy they would decay and step the loaded weights — and marks every row current after
                # it. (An in-place edit of the parameter's storage by other code cannot be intercepted: call flush() first.)
                def _loaded(*a, **k):
                    if self.deferred is not None:
                        self.deferred.mark_all_current()
                self._hooks = [net.register_state_dict_pre_hook(lambda *a, **k: self.flush()),
                               fe.register_forward_pre_hook(lambda *a, **k: self.flush())]
                for mod in (net, fe):
                    self._hooks.append(mod.register_load_state_dict_pre_hook(lambda *a, **k: self.flush()))
                    self._hooks.append(mod.register_load_state_dict_post_hook(_loaded))

    def draw(self, u_shape, i_shape):
        """Modality draws of one step (user side first, as in SingleBranchNet.forward). May be called from the loader
        thread ahead of time: the entities' generators are consumed in step order either way."""
        net = self.net
        du = net.user_embedding_module._sample_modalities(tuple(u_shape)) if net.is_user_sb_module else None
        di = net.item_embedding_module._sample_modalities(tuple(i_shape))
        return du, di

    # ---- the launches of forward + losses + backward (no host work besides ctypes calls) -------------------------------------
    def _fwd_bwd(self, u, i, lab, pu, pi, su, si, seed):
        self._phase1(u, i, lab, pu, pi, su, si, seed)
        return self._phase2()

    def _phase1(self, u, i, lab, pu, pi, su, si, seed):
        """Forward, losses, scorer backward and the USER side's backward: after this phase every user-side gradient is final,
        so a data-parallel run can start reducing that part of the flat gradient buffer (for a lookup user side the whole
        user embedding table) while phase 2 runs."""
        a, st = self.arena, ops.stream()
        B, N = i.shape
        a.reset()
        ur = self.user.forward(u, pu, (seed, 0), su)                 # [B, D]; seed: device int64[1] of this step
        ir = self.item.forward(i, pi, (seed, 1), si)                 # [B*N, D]; None: the trailing BatchNorm runs inside the scorer
        tail = self.item.tail if ir is None else None
        D = self.item.D
        rl = self.rec_loss
        if self.kind == 0:
            scale = 1.0 / (B * N) if rl.aggregator == 'mean' else 1.0
        elif self.kind == 1:
            scale = 1.0 / (B * (N - 1)) if rl.aggregator == 'mean' else 1.0
        else:
            scale = 1.0 / B if rl.aggregator == 'mean' else 1.0
        shift = math.log(rl.n_items / rl.neg_train) if (self.kind == 2 and rl.train_neg_strategy == 'uniform') else 0.0
        no_reg = self.user.reg_loss is None and self.item.reg_loss is None
        self._packed = None
        if (tail is not None and no_reg and B >= 1
                and ops.lib().sbr_bn_score_loss_supported(int(D), int(N))):
            # scorer forward, loss + dlogits and the first backward pass of the fused tail in ONE launch: the slot rows of a user
            # stay in registers between the logits and the backward statistics (csrc/fused_tail.hip: bn_score_loss_kernel)
            z, mean, rstd = tail
            bn = self.item.trailing
            need = int(ops.lib().sbr_bn_score_loss_workspace())
            if self._tail_ws is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError('the loss workspace must exist before a step is captured (run one plain step first)')
                self._tail_ws = torch.zeros((need + 7) // 8, device=z.device, dtype=torch.float64)
            loss = a.f64()
            dlog = a.f32(B, N)
            dU = a.f32(B, D)
            self._packed = a.f64(3)
            call('sbr_bn_score_loss_fwd_bwd', ptr(z), ptr(ur), ptr(mean), ptr(rstd), ptr(bn.weight), ptr(bn.bias), self.kind, ptr(lab),
                 scale, shift, None, ptr(dlog), ptr(dU), ptr(loss), ptr(self._packed), B, N, D, ptr(self.item._bn_ws(bn, D)),
                 ptr(self._tail_ws), self._tail_ws.numel() * 8, st)
            self.user.backward(dU, self.one32)
            self._p2 = (None, loss, (dlog, ur))
            return
        logits = a.f32(B, N)
        if tail is not None:
            z, mean, rstd = tail
            bn = self.item.trailing
            call('sbr_bn_score_fwd', ptr(z), ptr(ur), ptr(mean), ptr(rstd), ptr(bn.weight), ptr(bn.bias), ptr(logits), B, N, D, st)
        else:
            call('sbr_score_dot_fwd', ptr(ur), ptr(ir), ptr(logits), B, N, D, st)
        loss = a.f64()
        dlog = a.f32(B, N)
        self._packed = None
        if no_reg and B >= 1:
            # one launch: no zeroing launch in front, block partial sums added in a fixed order, and the packed (total, rec, reg)
            # scalars of the step written by the same kernel (no regularisation losses: total = rec)
            need = int(ops.lib().sbr_rec_loss_workspace(B))
            if self._loss_ws is None or self._loss_ws.numel() * 8 < need:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError('the loss workspace must exist before a step is captured (run one plain step first)')
                if self._loss_ws is not None:
                    # outgrown by a larger batch: steps captured earlier have the old address baked in (counter + partials), so the
                    # old block is retired — never handed back to the caching allocator — and those graphs are dropped
                    ops._WS_RETIRED.append(self._loss_ws)
                    self._graphs.clear()
                self._loss_ws = torch.zeros((max(need, int(ops.lib().sbr_rec_loss_workspace(1 << 16))) + 7) // 8, device=logits.device,
                                            dtype=torch.float64)
            self._packed = a.f64(3)
            call('sbr_rec_loss_fwd_bwd_ws', self.kind, ptr(logits), ptr(lab), B, N, scale, shift, ptr(loss), ptr(dlog),
                 ptr(self._packed), ptr(self._loss_ws), self._loss_ws.numel() * 8, st)
        else:
            call('sbr_rec_loss_fwd_bwd', self.kind, ptr(logits), ptr(lab), B, N, scale, shift, ptr(loss), ptr(dlog), st)
        dU = a.f32(B, D)
        if tail is not None:
            # dU and the BatchNorm column sums in one pass over the pre-BatchNorm rows; the item gradient dlog[s] * u[b] is
            # recomputed by the apply pass of phase 2 instead of being stored
            z, mean, rstd = tail
            bn = self.item.trailing
            call('sbr_bn_score_bwd_stats', ptr(dlog), ptr(ur), ptr(z), ptr(dU), B, N, D, ptr(bn.weight), ptr(bn.bias), ptr(mean),
                 ptr(rstd), ptr(self.item._bn_ws(bn, D)), st)
            dI = None
        else:
            dI = a.f32(B * N, D)
            call('sbr_score_dot_bwd', ptr(dlog), ptr(ur), ptr(ir), ptr(dU), ptr(dI), B, N, D, st)
        self.user.backward(dU, self.one32)
        self._p2 = (dI, loss, (dlog, ur) if tail is not None else None)

    def _phase2(self):
        """The item side's backward (the bulk of the step) and the loss scalars."""
        a, st = self.arena, ops.stream()
        dI, loss, tail = self._p2
        self.item.backward(dI, self.one32, tail)
        if self._packed is not None:
            return self._packed
        out = a.f64(3)                                               # (total, rec, reg)
        ru, ri = self.user.reg_loss, self.item.reg_loss
        call('sbr_pack_losses', ptr(loss), ptr(ru), float(getattr(self.user, 'reg_w', 0.0)), ptr(ri),
             float(getattr(self.item, 'reg_w', 0.0)), ptr(out), st)
        return out

    # ---- host side of a step: draw, plan, uploads (may run on the loader thread) ---------------------------------------------
    def prepare(self, u_idxs, i_idxs, labels, draws=None, ahead: bool = True, labels_key=None) -> 'PreparedBatch':
        """Everything of a step that is not a kernel launch: modality draw (entity generators, consumed in step order),
        counting-sort plan, and the upload of indices / labels / slot lists. With ``ahead`` (loader thread) the uploads go
        through pinned staging buffers on a private stream and the step only waits on their event, so the launch thread never
        blocks on PCIe. The index buffers carry one sentinel entry behind them (= their first entry): the entity that the
        padded launches of a graph-mode plan resolve (``_EntityRun.plan``).
        ``labels_key``: a hashable promise by the caller that every batch passed with this key and this shape carries the
        SAME label matrix (the default loader: first column 1, negatives 0) — it is then uploaded once and kept on the
        device instead of travelling with every batch."""
        dev = torch.device(self.net.device)
        u, i, lab = torch.as_tensor(u_idxs).long(), torch.as_tensor(i_idxs).long(), torch.as_tensor(labels).double()
        du, di = draws if draws is not None else self.draw(u.shape, i.shape)
        pad = self.use_graph          # also while KernelTimer forces plain launches: the timed launches keep the graph's shapes
        pb = PreparedBatch()
        pb.u_shape, pb.i_shape = tuple(u.shape), tuple(i.shape)
        pb.pu, pb.pi = self.user.plan(du, pad), self.item.plan(di, pad)
        pb.packed = pb.layout = None
        pb.native = None             # (NativeBatchProducer, slot) for batches of the native producer (native_loader.py)
        self._n_prepared += 1                                        # dropout seed of this step (travels with the batch)
        seed = (torch.initial_seed() * 1000003 + 2 * self._n_prepared) & 0x3FFFFFFFFFFFFFFF
        pb.lab_cached = False
        cached_lab = self._label_cache.get((labels_key, tuple(lab.shape))) if labels_key is not None else None
        if ahead and self._up_stream is None:
            self._up_stream = torch.cuda.Stream(dev, priority=-1)    # the loader thread's upload stream
        stream = self._up_stream if ahead else torch.cuda.current_stream(dev)
        parts = [u.reshape(-1), i.reshape(-1), lab.reshape(-1)]
        with torch.cuda.device(dev), torch.cuda.stream(stream):
            if all(t.device.type == 'cpu' for t in parts):
                # one packed H2D copy: [u | u[0] | i | i[0] | labels | user modality draw | item modality draw | dropout seed],
                # 16-byte aligned segments
                un, inn = parts[0].numpy(), parts[1].numpy()
                arrs = [np.concatenate([un, un[:1]]), np.concatenate([inn, inn[:1]]),
                        parts[2].numpy() if cached_lab is None else np.empty(0, np.float64),
                        pb.pu[0] if pb.pu is not None else np.empty(0, np.int8), pb.pi[0], np.array([seed], dtype=np.int64)]
                offs = [0]
                for a_ in arrs:
                    offs.append((offs[-1] + a_.nbytes + 15) & ~15)
                if ahead:
                    host, ev = self._staging(offs[-1])
                else:
                    host, ev = torch.empty(max(offs[-1], 16), dtype=torch.uint8), None
                hv = host.numpy()
                for a_, o in zip(arrs, offs):
                    hv[o:o + a_.nbytes] = np.ascontiguousarray(a_).view(np.uint8).reshape(-1)
                packed = to_device(host[:offs[-1]], dev)             # pinned ring slot: asynchronous; pageable: synchronous
                if ev is not None:
                    ev.record(stream)                                # the staging slot is free again after this copy
                pb.packed, pb.layout = packed, tuple((o, a_.nbytes) for a_, o in zip(arrs, offs))
                pb.u, pb.i, pb.lab, pb.su, pb.si, pb.seed = self._views(packed, pb.layout, pb.pu is not None)
                if cached_lab is not None:
                    pb.lab, pb.lab_cached = cached_lab, True
                elif labels_key is not None:
                    keep = pb.lab.clone()                              # outlives this batch's packed buffer
                    self._label_cache[(labels_key, tuple(lab.shape))] = keep
            else:                                                     # indices already on the device
                def ext(t):
                    t = to_device(t, dev)
                    return torch.cat([t, t[:1]])
                pb.u, pb.i, pb.lab = ext(parts[0]), ext(parts[1]), to_device(parts[2], dev).contiguous()
                pb.su = to_device(torch.from_numpy(pb.pu[0]), dev) if pb.pu is not None else None
                pb.si = to_device(torch.from_numpy(pb.pi[0]), dev)
                pb.seed = torch.tensor([seed], dtype=torch.int64, device=dev)
            pb.event = None
            if ahead:
                pb.event = torch.cuda.Event()
                pb.event.record(stream)
        return pb

    @staticmethod
    def _views(packed, layout, has_su):
        v = [packed[o:o + n].view(dt) for (o, n), dt in
             zip(layout, (torch.int64, torch.int64, torch.float64, torch.int8, torch.int8, torch.int64))]
        if not has_su:
            v[3] = None
        return v

    def _staging(self, nbytes):
        """Next slot of the pinned staging ring (hipHostMalloc once per slot — per-batch pin_memory() costs milliseconds)."""
        if not hasattr(self, '_pin'):
            self._pin, self._pin_next = [None] * 8, 0
        k = self._pin_next
        self._pin_next = (k + 1) % len(self._pin)
        slot = self._pin[k]
        if slot is None or slot[0].numel() < nbytes:
            slot = self._pin[k] = (torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, pin_memory=True), torch.cuda.Event())
        else:
            slot[1].synchronize()                                     # its previous upload has left the buffer
        return slot

    def _capture(self, key, pb):
        cs = _CapturedStep()
        if pb.packed is not None:                                    # static copy of the packed upload: one D2D per replay
            cs.packed = torch.empty_like(pb.packed)
            cs.u, cs.i, cs.lab, cs.su, cs.si, cs.seed = self._views(cs.packed, pb.layout, pb.su is not None)
            if pb.lab_cached:
                cs.lab = pb.lab                                      # device-resident constant labels: nothing to copy
        else:
            cs.packed = None
            cs.u, cs.i, cs.lab = torch.empty_like(pb.u), torch.empty_like(pb.i), torch.empty_like(pb.lab)
            cs.su = torch.empty_like(pb.su) if pb.su is not None else None
            cs.si = torch.empty_like(pb.si)
            cs.seed = torch.empty_like(pb.seed)
        cs.graph = torch.cuda.CUDAGraph()
        torch.cuda.current_stream().synchronize()
        # thread_local: the loader thread keeps issuing its own copies / kernels on its streams meanwhile
        cs.graph2 = None
        with torch.cuda.graph(cs.graph, capture_error_mode='thread_local'):
            with pin_stream():
                self._phase1(cs.u[:-1].view(pb.u_shape), cs.i[:-1].view(pb.i_shape), cs.lab, pb.pu, pb.pi, cs.su, cs.si, cs.seed)
                if not self.split:
                    cs.out = self._phase2()
        if self.split:                                               # second graph, same memory pool, replayed in order
            cs.graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cs.graph2, pool=cs.graph.pool(), capture_error_mode='thread_local'):
                with pin_stream():
                    cs.out = self._phase2()
        self._graphs[key] = cs
        return cs

    def step(self, u_idxs, i_idxs, labels, draws=None):
        """One training step. Returns device tensors (total loss f64, rec loss f64, reg loss f64) — no host sync.
        ``draws``: None, the result of ``draw()``, or a ``PreparedBatch`` made ahead of time by ``prepare()``."""
        net = self.net
        if not net.training:
            raise RuntimeError('FusedTrainStep.step() needs the model in train mode')
        with pin_stream():
            pb = draws if isinstance(draws, PreparedBatch) else self.prepare(u_idxs, i_idxs, labels, draws, ahead=False)
            native = getattr(pb, 'native', None)
            if native is not None:
                native[0].wait(native[1])                            # this stream waits for the producer's upload of the slot
            if pb.event is not None:
                cur = torch.cuda.current_stream()
                cur.wait_event(pb.event)
                for t in (pb.packed, pb.u, pb.i, pb.lab, pb.su, pb.si, pb.seed):   # allocated on the upload stream, consumed here
                    if t is not None:
                        t.record_stream(cur)
            self.n_steps += 1
            self.last_out3 = None
            out = None
            if self._sparse is None:
                self._setup_sparse_exchange(int(pb.u_shape[0]))
                if self._sparse is False and self.deferred is not None:       # dense all-reduce: touched rows unknown
                    self._drop_deferred()
            if self.deferred is not None:
                self.deferred.catch_up(pb.u[:-1])                            # the rows this step's forward pass reads
            if self.use_graph and pb.pi[5] and not ops.KernelTimer.enabled:
                if self._arena_buf is not self.arena.buf:            # the arena moved: captured addresses are stale
                    self._graphs.clear()
                    self._arena_buf = self.arena.buf
                key = (pb.u_shape, pb.i_shape, pb.pu[1:] if pb.pu is not None else None, pb.pi[1:], pb.lab_cached)
                cs = self._graphs.get(key, 0)
                if cs is None and self.arena.high <= self.arena.buf.numel():
                    cs = self._capture(key, pb)                      # second sighting: the arena is sized, capture
                elif cs == 0 and len(self._graphs) < self.MAX_GRAPHS:
                    self._graphs[key] = None                         # first sighting: plain launches (sizes the arena)
                if isinstance(cs, _CapturedStep):
                    if cs.packed is not None and pb.packed is not None:
                        cs.packed.copy_(pb.packed, non_blocking=True)
                    else:
                        cs.u.copy_(pb.u, non_blocking=True)
                        cs.i.copy_(pb.i, non_blocking=True)
                        cs.lab.copy_(pb.lab, non_blocking=True)
                        if cs.su is not None:
                            cs.su.copy_(pb.su, non_blocking=True)
                        cs.si.copy_(pb.si, non_blocking=True)
                        cs.seed.copy_(pb.seed, non_blocking=True)
                    cs.graph.replay()
                    pending = self._reduce_user_part()
                    if cs.graph2 is not None:
                        cs.graph2.replay()
                    self.n_replays += 1
                    out = cs.out                                     # the graph's static buffer: copied out below
            if out is None:
                self._phase1(pb.u[:-1].view(pb.u_shape), pb.i[:-1].view(pb.i_shape), pb.lab, pb.pu, pb.pi, pb.su, pb.si, pb.seed)
                pending = self._reduce_user_part()
                out = self._phase2().clone().unbind(0)
            # ---- reduce + update
            self._reduce_rest(pending)
            # step() + zero_grad() in the optimizer's own launch — which also carries the step's loss scalars out of the captured
            # step's static buffer into a fresh tensor (instead of a clone launch)
            static = out if torch.is_tensor(out) else None
            fresh = torch.empty_like(static) if static is not None else None
            cp = (static, fresh) if static is not None else None
            # with a deferred lookup table the same launch updates it row by row: only the rows that received gradient are touched
            rows = None if self.deferred is None else (self._touched_rows if self._sparse else pb.u[:-1])
            if self.deferred is not None and self._sparse:
                self.deferred.sweep_ok = False                       # rows of other ranks' batches: not claimed by this rank's catch-up
            took = self.opt.step_flat(zero_grad=True, copy=cp, rows=rows)
            if static is not None:
                self.last_out3 = fresh if took else static.clone()   # (total, rec, reg) of this step as one [3] tensor
                out = self.last_out3.unbind(0)
            if native is not None:
                native[0].release(native[1])                         # everything queued so far has read the slot
            return out

    def flush(self):
        """Bring a row-wise updated table up to date (no-op otherwise). Called automatically before state_dict(), the user
        module's own forward, a dense optimizer step and close(); call it before reading parameter tensors directly."""
        if self.deferred is not None:
            with pin_stream():
                self.deferred.flush()

    def _drop_deferred(self):
        self.flush()
        for h in getattr(self, '_hooks', []):
            h.remove()
        self.opt.deferred = None
        self.deferred = None

    # ---- data-parallel gradient exchange, overlapped with the item side's backward -----------------------------------------------
    def _user_range(self):
        """[lo, hi) of the user side's parameters in the flat buffers, or None when they are not one contiguous run."""
        fp = self.opt.fp
        ids = {id(p) for p in self.net.user_embedding_module.parameters()}
        idx = [k for k, p in enumerate(fp.params) if id(p) in ids]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            return None
        hi = fp.offsets[idx[-1] + 1] if idx[-1] + 1 < len(fp.offsets) else fp.total
        return fp.offsets[idx[0]], hi

    def _setup_sparse_exchange(self, B: int):
        """First step of a data-parallel run (one host sync): the ranks agree on the row capacity of the exchange — the largest
        first batch — and on whether the sparse exchange moves fewer bytes than the dense all-reduce of the table gradient."""
        import torch.distributed as dist
        dev = torch.device(self.net.device)
        cap = torch.tensor([B], device=dev, dtype=torch.int64)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        cap = int(cap.item())
        world = dist.get_world_size()
        table = self.user.fe.front_params()[0]
        U, D = table.shape
        # all-gather: every rank receives world * cap * (D + 1) words; ring all-reduce: about 2 * U * D words
        if world * cap * (D + 1) >= 2 * U * D:
            self._sparse = False
            return
        send = torch.zeros(cap * (D + 1), device=dev, dtype=torch.float32)
        recv = torch.zeros(world, cap * (D + 1), device=dev, dtype=torch.float32)
        self._sparse = (send, recv, cap)
        self.user.xchg = (send[:cap * D].view(cap, D), send[cap * D:].view(torch.int32))

    def _reduce_user_part(self):
        """After phase 1: start the exchange of the user side's gradients on RCCL's stream (asynchronous: the launches of
        phase 2 go to the compute stream right behind). Returns the pending work handle (None when not distributed)."""
        if not (self.split and parallel.is_distributed()) or self._urange is None:
            return None
        if self._sparse:
            import torch.distributed as dist
            send, recv, _ = self._sparse
            return dist.all_gather_into_tensor(recv.view(-1), send, async_op=True)
        lo, hi = self._urange
        return parallel.all_reduce_async(self.opt.fp.grad[lo:hi])

    def _scatter_user_rows(self):
        """Sparse exchange, after the all-gather: every rank adds every rank's (row, gradient) pairs into its dense table
        gradient — the sum the dense all-reduce would have produced. The gathered lists are identical on every rank and are
        added in an order fixed by a stable sort by table row (no float atomics), so the replicas stay bit-identical."""
        _, recv, cap = self._sparse
        table = self.user.fe.front_params()[0]
        dW = _grad_of(table)
        D = table.shape[1]
        W = recv.shape[0]
        rows = recv[:, cap * D:].view(torch.int32).reshape(-1)        # [W * cap], rank-major (a copy: the view is strided)
        self._touched_rows = rows                                     # every row that receives gradient this step
        rows_sorted, perm = torch.sort(rows, stable=True)
        call('sbr_scatter_add_rows_sorted', ptr(recv), D, cap, recv.stride(0), ptr(perm), ptr(rows_sorted), ptr(dW),
             dW.stride(0), W * cap, D, ops.stream())

    def _reduce_rest(self, pending):
        if not parallel.is_distributed():
            return
        g = self.opt.fp.grad
        if pending is None:
            if self._sparse:
                raise RuntimeError('sparse user exchange without the two-phase step')
            parallel.all_reduce_flat_(g)
            return
        lo, hi = self._urange
        parts = [g[:lo], g[hi:]]
        works = [parallel.all_reduce_async(t) for t in parts if t.numel()]
        for w in works + [pending]:
            w.wait()                                                  # the compute stream waits; the host does not block
        if self._sparse:
            self._scatter_user_rows()
        g.div_(parallel.world_size())

    def check_errors(self):
        """Host check of the sticky missing-id flags (one device sync): raises KeyError like the reference's feature lookup."""
        self.net.check_index_errors()

    def close(self):
        """Drop the captured graphs (also done at interpreter exit: hipGraph objects must not outlive the HIP runtime) and bring
        a row-wise updated table up to date."""
        self.flush()
        self._graphs.clear()


class PreparedBatch:
    """Device-resident inputs of one step + its launch plan (``FusedTrainStep.prepare``)."""
    __slots__ = ('packed', 'layout', 'u', 'i', 'lab', 'su', 'si', 'seed', 'pu', 'pi', 'u_shape', 'i_shape', 'event', 'lab_cached', 'native')


_LIVE = weakref.WeakSet()


@atexit.register
def _drop_graphs():
    for f in list(_LIVE):
        try:
            f.close()
        except Exception:          # interpreter teardown: the HIP runtime may already be gone
            f._graphs.clear()


class _CapturedStep:
    __slots__ = ('graph', 'graph2', 'packed', 'u', 'i', 'lab', 'su', 'si', 'seed', 'out')
