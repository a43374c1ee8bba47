"""Fused training step: the whole hot loop body of train/trainer.py:204-223 as ONE straight-line sequence of HIP kernel
launches — forward, losses, hand-written backward, gradient all-reduce, fused dense optimizer — without the autograd
tape, without per-op tensor allocation and without a host sync.

It computes exactly what ``loss(model(u, i)).backward(); optimizer.step()`` computes through the ``nn.Module`` surface
(``tests/test_hip_golden.py::test_fused_step_matches_autograd_path`` checks parameters after several steps): the modules
keep owning the parameters (flat buffer of ``optim.FlatParameters``), this class only replaces the launch choreography.
Per step the host does: the modality draw + counting sort (numpy), three index uploads, ~120 ctypes calls.

Supported (everything the shipped sbnet configurations use): user side = embedding lookup / any plain FeatureEmbedding or a
SingleBranchNetEntity; item side = SingleBranchNetEntity; rec losses bce / bpr / sampled softmax; InfoNCE regularisation.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import ops, parallel
from ._lib import call, pin_stream, ptr
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

    # ---- forward -----------------------------------------------------------------------------------------------------
    def forward(self, idx: torch.Tensor, draw: Tuple[np.ndarray, list], seed: int):
        ent, a, st = self.ent, self.a, ops.stream()
        pos, order = draw
        k = pos.shape[1]
        flat = pos.reshape(-1)
        R = flat.size
        order_idx = np.argsort(flat, kind='stable').astype(np.int32)
        counts = np.bincount(flat, minlength=len(order))
        dev = idx.device
        slots = torch.from_numpy(order_idx).to(dev, non_blocking=True)
        entries, tables, offs = [], [], [0]
        for m, c in enumerate(counts.tolist()):
            if c:
                fe = ent.modality_modules[order[m]]
                entries.append((fe, offs[-1], c))
                tables.append(fe._table)
                offs.append(offs[-1] + c)
        idx_flat = idx.reshape(-1)
        rows, _ = resolve_rows(idx_flat, k, slots, offs, tables)
        self.entries, self.rows, self.slots, self.R, self.k, self.shape = entries, rows, slots, R, k, tuple(idx.shape)
        x0 = a.f32(R, self.C)
        self.hidden = [fe.front_forward(fe.front_params(), rows[o:o + n], n, x0, slots[o:o + n]) for fe, o, n in entries]
        self.x0 = x = x0
        if self.normalize:
            xn, self.inv = a.f32(R, self.C), a.f32(R)
            call('sbr_l2norm_fwd', ptr(x), ptr(xn), ptr(self.inv), R, self.C, ops.NORM_EPS, st)
            self.xn = x = xn
        self.seed = None
        if self.p_drop:
            xd = a.f32(R, self.C)
            call('sbr_dropout', ptr(x), ptr(xd), x.numel(), float(self.p_drop), seed, st)
            self.seed, x = seed, xd
        self.acts = []                                   # per layer: (input, pre-BN output | None, output, mean, rstd)
        for lin, bn, act in self.layers:
            w = lin.weight
            if bn is None:
                y = ops.linear_nt(x, w, lin.bias, act, out=a.f32(R, w.shape[0]))
                self.acts.append((x, None, y, None, None))
            else:
                z = ops.linear_nt(x, w, lin.bias, 0, out=a.f32(R, w.shape[0]))
                y, mean, rstd = self._bn_fwd(bn, z, act)
                self.acts.append((x, z, y, mean, rstd))
            x = y
        self.tb = None
        if self.trailing is not None:
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
            if N > ops.infonce_max_n():
                raise NotImplementedError(f'InfoNCE over {N} rows per group exceeds the on-chip kernel limit')
            self.G, self.N = G, N
            self.reg_loss = a.f64()
            e3 = x.view(S, 2, self.D)
            call('sbr_infonce_fwd', e3[:, 0].data_ptr(), e3[:, 1].data_ptr(), 2 * self.D, G, N, self.D, self.tau,
                 1.0 / (G * N), ptr(self.reg_loss), st)
        if k == 1:
            return x.view(S, self.D)
        out = a.f32(S, self.D)
        self.arg = a.u8(S, self.D) if ent._agg_mode == 1 else None
        call('sbr_aggregate_fwd', ptr(x), ptr(out), ptr(self.arg), S, k, self.D, ent._agg_mode, st)
        return out

    def _bn_fwd(self, bn, x, act):
        a, st = self.a, ops.stream()
        n, D = x.shape
        y, mean, rstd, ws = a.f32(n, D), a.f32(D), a.f32(D), a.f64(2 * D)
        call('sbr_bn_train_fwd', ptr(x), ptr(y), n, D, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
             ptr(bn.num_batches_tracked), ptr(mean), ptr(rstd), ptr(ws), ops.BN_EPS, ops.BN_MOMENTUM, act, st)
        return y, mean, rstd

    def _bn_bwd(self, bn, dy, y, x, mean, rstd, act):
        a, st = self.a, ops.stream()
        n, D = x.shape
        dx, ws = a.f32(n, D), a.f64(2 * D)
        call('sbr_bn_train_bwd', ptr(dy), ptr(y), ptr(x), ptr(dx), n, D, ptr(bn.weight), ptr(mean), ptr(rstd),
             ptr(_grad_of(bn.weight)), ptr(_grad_of(bn.bias)), ptr(ws), act, st)
        return dx

    # ---- backward ------------------------------------------------------------------------------------------------------
    def backward(self, dout: torch.Tensor, one_f32: torch.Tensor):
        ent, a, st = self.ent, self.a, ops.stream()
        R, k, D = self.R, self.k, self.D
        S = R // k
        if k == 1:
            de = dout
        else:
            de = a.f32(R, D)
            call('sbr_aggregate_bwd', ptr(dout), ptr(self.arg), ptr(de), S, k, D, ent._agg_mode, st)
        if self.reg:
            dreg = a.f32(R, D)
            e3, d3 = self.e.view(S, 2, D), dreg.view(S, 2, D)
            # d(total)/d(reg_loss) = regularization_weight (sgd_alg.py:2002); mean over G*N rows inside the kernel
            call('sbr_infonce_bwd', e3[:, 0].data_ptr(), e3[:, 1].data_ptr(), 2 * D, self.G, self.N, D, self.tau,
                 self.reg_w / (self.G * self.N), ptr(one_f32), d3[:, 0].data_ptr(), d3[:, 1].data_ptr(), 2 * D, st)
            de = de.add_(dreg) if k > 1 else dreg.add_(de)
        d = de
        if self.tb is not None:
            x, y, mean, rstd = self.tb
            d = self._bn_bwd(self.trailing, d, y, x, mean, rstd, 0)
        for (lin, bn, act), (x, z, y, mean, rstd) in zip(reversed(self.layers), reversed(self.acts)):
            if bn is not None:
                dz = self._bn_bwd(bn, d, y, z, mean, rstd, act)
            else:
                dz = ops.act_grad(d, y, act) if act else d
            w = lin.weight
            ops.matmul_tn(dz, x, out=_grad_of(w))
            ops.colsum(dz, out=_grad_of(lin.bias))
            d = ops.matmul_nn(dz, w)
        if self.seed is not None:
            dd = a.f32(R, self.C)
            call('sbr_dropout', ptr(d), ptr(dd), d.numel(), float(self.p_drop), self.seed, st)
            d = dd
        if self.normalize:
            dn = a.f32(R, self.C)
            call('sbr_l2norm_bwd', ptr(d), ptr(self.xn), ptr(self.inv), ptr(dn), R, self.C, ops.NORM_EPS, st)
            d = dn
        for (fe, o, n), hs in zip(self.entries, self.hidden):
            ps = fe.front_params()
            fe.front_backward(ps, hs, self.rows[o:o + n], n, self.x0, d, self.slots[o:o + n], grad_out=[_grad_of(p) for p in ps])


class _PlainRun:
    """Forward / backward of a plain FeatureEmbedding side (embedding lookup or projector on one feature)."""

    def __init__(self, fe: FeatureEmbedding, arena: Arena):
        if fe.post_embedding_layers is not None:
            raise NotImplementedError('post_embedding_layers are not part of the SingleBranchNet path')
        self.fe, self.a = fe, arena
        self.reg_loss = None

    def forward(self, idx, draw=None, seed=0):
        fe, a = self.fe, self.a
        flat = idx.reshape(-1)
        n = flat.numel()
        self.n = n
        self.slots = torch.arange(n, device=flat.device, dtype=torch.int32)
        self.rows, _ = resolve_rows(flat, 1, self.slots, [0, n], [fe._table])
        self.out = a.f32(n, fe.front_dim)
        self.hidden = fe.front_forward(fe.front_params(), self.rows, n, self.out, None)
        return self.out

    def backward(self, dout, one_f32):
        fe = self.fe
        ps = fe.front_params()
        fe.front_backward(ps, self.hidden, self.rows, self.n, self.out, dout, None, grad_out=[_grad_of(p) for p in ps])


class FusedTrainStep:
    def __init__(self, net: SingleBranchNet, rec_loss, optimizer):
        if not isinstance(net.item_embedding_module, SingleBranchNetEntity):
            raise NotImplementedError('FusedTrainStep needs a SingleBranchNetEntity item side')
        self.net, self.rec_loss, self.opt = net, rec_loss, optimizer
        dev = net.device
        self.arena = Arena(dev)
        self.user = (_EntityRun if net.is_user_sb_module else _PlainRun)(net.user_embedding_module, self.arena)
        self.item = _EntityRun(net.item_embedding_module, self.arena)
        self.kind = {RecBinaryCrossEntropy: 0, RecBayesianPersonalizedRankingLoss: 1, RecSampledSoftmaxLoss: 2}[type(rec_loss)]
        self.one64 = torch.ones((), device=dev, dtype=torch.float64)
        self.one32 = torch.ones((), device=dev, dtype=torch.float32)
        self.n_steps = 0
        self.opt.zero_grad()

    def draw(self, u_shape, i_shape):
        """Modality draws of one step (user side first, as in SingleBranchNet.forward). May be called from the loader
        thread ahead of time: the entities' generators are consumed in step order either way."""
        net = self.net
        du = net.user_embedding_module._sample_modalities(tuple(u_shape)) if net.is_user_sb_module else None
        di = net.item_embedding_module._sample_modalities(tuple(i_shape))
        return du, di

    def step(self, u_idxs, i_idxs, labels, draws=None):
        """One training step. Returns device tensors (total loss f64, rec loss f64, reg loss f64) — no host sync."""
        net, a = self.net, self.arena
        dev = net.device
        if not net.training:
            raise RuntimeError('FusedTrainStep.step() needs the model in train mode')
        with pin_stream() as st:
            u = u_idxs.to(dev, non_blocking=True).long().contiguous()
            i = i_idxs.to(dev, non_blocking=True).long().contiguous()
            lab = labels.to(dev, non_blocking=True).double().contiguous()
            B, N = i.shape
            du, di = draws if draws is not None else self.draw(u.shape, i.shape)
            a.reset()
            self.n_steps += 1
            seed = (torch.initial_seed() * 1000003 + 2 * self.n_steps) & 0x7FFFFFFFFFFFFFFF
            # ---- forward
            ur = self.user.forward(u, du, seed)                      # [B, D]
            ir = self.item.forward(i, di, seed + 1)                  # [B*N, D]
            D = ir.shape[-1]
            logits = a.f32(B, N)
            call('sbr_score_dot_fwd', ptr(ur), ptr(ir), ptr(logits), B, N, D, st)
            rl = self.rec_loss
            if self.kind == 0:
                scale = 1.0 / (B * N) if rl.aggregator == 'mean' else 1.0
            elif self.kind == 1:
                scale = 1.0 / (B * (N - 1)) if rl.aggregator == 'mean' else 1.0
            else:
                scale = 1.0 / B if rl.aggregator == 'mean' else 1.0
            import math
            shift = math.log(rl.n_items / rl.neg_train) if (self.kind == 2 and rl.train_neg_strategy == 'uniform') else 0.0
            loss = a.f64()
            call('sbr_rec_loss_fwd', self.kind, ptr(logits), ptr(lab), B, N, scale, shift, ptr(loss), st)
            # ---- backward
            dlog = a.f32(B, N)
            call('sbr_rec_loss_bwd', self.kind, ptr(logits), ptr(lab), B, N, scale, shift, ptr(self.one64), 1, ptr(dlog), st)
            dU, dI = a.f32(B, D), a.f32(B * N, D)
            call('sbr_score_dot_bwd', ptr(dlog), ptr(ur), ptr(ir), ptr(dU), ptr(dI), B, N, D, st)
            self.item.backward(dI, self.one32)
            self.user.backward(dU, self.one32)
            # ---- reduce + update
            if parallel.is_distributed():
                parallel.all_reduce_flat_(self.opt.fp.grad)
            self.opt.step_flat()
            self.opt.fp.grad.zero_()
            reg = torch.zeros((), device=dev, dtype=torch.float64)
            for side in (self.user, self.item):
                if side.reg_loss is not None:
                    reg = reg + side.reg_loss * side.reg_w
            rec = loss.clone()
            return rec + reg, rec, reg
